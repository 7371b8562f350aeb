#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_25; mkdir -p $O
( timeout 900 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3 ) > $O/tests.log 2>&1
for i in 1 2 3; do bash tools/ab_train.sh main f4; done > $O/ab.log 2>&1
cat $O/tests.log | cut -c1-200; cat $O/ab.log
