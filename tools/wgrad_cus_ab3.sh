#!/bin/bash
# GPU box: very small batches - side-stream count under the CU cap (interleaved A/B on one box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
run() {  # label, rays, env...
  local label=$1 rays=$2; shift 2
  env "$@" python bench.py --train --train-rays $rays --steps 40 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label rays $rays: %.3f ms, full objective %.3f' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step', float('nan'))))"
}
for rep in 1 2 3; do
  for rays in 128 256 768; do
    run "nocap/1-stream" $rays NERFDS_TRAIN_WGRAD_CUS=0 NERFDS_TRAIN_SIDE_SMALL=100000
    run "cap128/1-stream" $rays NERFDS_TRAIN_SIDE_SMALL=100000
    run "cap128/3-streams" $rays NERFDS_TRAIN_SIDE_SMALL=0
    run "cap64/3-streams" $rays NERFDS_TRAIN_SIDE_SMALL=0 NERFDS_TRAIN_WGRAD_CUS=64
  done
done | tee $OUT/wgrad_cus_ab3.txt
