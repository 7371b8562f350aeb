#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_9; mkdir -p $O
( timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py tests/test_rccl_single_gpu.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED" | head -40 ) > $O/train_tests.log 2>&1
for m in 1 0 1 0; do echo -n "merged=$m: "; NERFDS_TRAIN_MERGED=$m python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'])"; done > $O/ab_train_merged.txt 2>&1
cat $O/train_tests.log $O/ab_train_merged.txt
