#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t1
timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py tests/test_train_gemm.py -m gpu -q 2>&1 | grep -E "^E   Assert|^E       assert|passed|failed|^FAILED" | head -40 > gpurun_out/r3t1/tests.log
cat gpurun_out/r3t1/tests.log
