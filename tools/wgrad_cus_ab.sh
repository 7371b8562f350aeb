#!/bin/bash
# GPU box: the training step with the side streams' persistent weight-gradient workgroups capped at N CUs (NERFDS_TRAIN_WGRAD_CUS), interleaved A/B
# usage: tools/wgrad_cus_ab.sh "<caps>" "<rays>" [reps]
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
CAPS=${1:-"0 240 224 192 160 128"}; RAYS=${2:-"4096 512"}; REPS=${3:-3}
for rep in $(seq 1 $REPS); do
  for c in $CAPS; do
    for rays in $RAYS; do
      NERFDS_TRAIN_WGRAD_CUS=$c python bench.py --train --train-rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $c rays $rays rep $rep: %.3f ms, full objective %s' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step')))"
    done
  done
done | tee -a $OUT/wgrad_cus_ab.txt
