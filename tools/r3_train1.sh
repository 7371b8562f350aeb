#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t1
timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py tests/test_train_gemm.py -m gpu -q 2>&1 | grep -E "^E   Assert|passed|failed|^FAILED" | head -20 > gpurun_out/r3t1/tests.log
timeout 600 python bench.py --train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'], d['roofline']['frac'])" > gpurun_out/r3t1/bench.log
cat gpurun_out/r3t1/tests.log gpurun_out/r3t1/bench.log
