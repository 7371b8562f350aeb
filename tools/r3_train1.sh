#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t1
timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -s 2>&1 | grep -E "worst|target_norm|passed|failed|^FAILED" > gpurun_out/r3t1/measured.log
cat gpurun_out/r3t1/measured.log
