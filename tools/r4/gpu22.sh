#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_22; mkdir -p $O
( timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log 2>&1
for i in 1 2; do bash tools/ab_train.sh main s4; done > $O/ab.log 2>&1
timeout 600 bash tools/train_timeline.sh r4f > /dev/null 2>&1
cat $O/tests.log | cut -c1-250; cat $O/ab.log; grep "train_backward" gpurun_out/timeline_r4f.txt | cut -c1-110
