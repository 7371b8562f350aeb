#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_train.sh r4b_train > /dev/null 2>&1
bash tools/train_timeline.sh r4b > /dev/null 2>&1
for n in 1 2 3 4; do NERFDS_TRAIN_SIDE_STREAMS=$n python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side $n', d['ms_per_step'])"; done > gpurun_out/r4b_side.log
head -34 gpurun_out/prof_r4b_train/summary.txt | cut -c1-170; cat gpurun_out/r4b_side.log
