#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_21; mkdir -p $O
( timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log 2>&1
for i in 1 2 3; do
timeout 300 python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['loss_last'])" >> $O/bench.log 2>&1
done
timeout 600 bash tools/train_timeline.sh r4e > /dev/null 2>&1
cat $O/tests.log | cut -c1-250; cat $O/bench.log; grep "composite\|bott\|small_gemm" gpurun_out/timeline_r4e.txt | cut -c1-100
