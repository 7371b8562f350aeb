#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_10; mkdir -p $O
for rep in 1 2; do for n in 2 3 4 6; do echo -n "side=$n: "; NERFDS_TRAIN_SIDE_STREAMS=$n python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_last'])"; done; done > $O/ab_side.txt 2>&1
cat $O/ab_side.txt
