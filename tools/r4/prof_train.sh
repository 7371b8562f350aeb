#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_train.sh r4_train > /dev/null 2>&1
bash tools/prof_train_traffic.sh r4_train_traffic > /dev/null 2>&1
bash tools/train_timeline.sh r4 > /dev/null 2>&1
python bench.py --train > gpurun_out/r4_bench_train.json 2>/dev/null
head -30 gpurun_out/prof_r4_train/summary.txt | cut -c1-170; head -8 gpurun_out/prof_r4_train_traffic/traffic.json; tail -c 1200 gpurun_out/r4_bench_train.json
