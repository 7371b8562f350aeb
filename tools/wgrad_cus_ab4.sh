#!/bin/bash
# GPU box: round 6's final launch rules of the side streams (CU cap of the persistent weight-gradient launches, stream count) against the rules before them
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
run() {  # label, rays, env...
  local label=$1 rays=$2; shift 2
  env "$@" python bench.py --train --train-rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label rays $rays: %.3f ms, full objective %.3f' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step', float('nan'))))"
}
for rep in 1 2 3; do
  for rays in 128 512 1024 2048 4096; do
    run "before (no cap, one stream up to 768 rays)" $rays NERFDS_TRAIN_WGRAD_CUS=0 NERFDS_TRAIN_SIDE_SMALL=768
    run "after  (defaults)" $rays X=1
  done
done | tee $OUT/wgrad_cus_ab4.txt
