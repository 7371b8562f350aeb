#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_18; mkdir -p $O
for i in 1 2; do bash tools/ab_train.sh main w4; done > $O/ab.log 2>&1
for n in 0 1 3; do export NERFDS_TRAIN_SIDE_STREAMS=$n; echo "side $n"; bash tools/ab_train.sh main; done >> $O/ab.log 2>&1
cat $O/ab.log
