#!/bin/bash
# GPU box: k_fold_rgb with 16 loads in flight against the rolled loop (measurement variant built by tools/variant.sh 'foldold:train_k:-DNERFDS_EXP_FOLD_ROLLED=1'), interleaved
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_foldold.so
run() {  # label, rays, env...
  local label=$1 rays=$2; shift 2
  env "$@" python bench.py --train --train-rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label rays $rays: %.3f ms, full objective %.3f' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step', float('nan'))))"
}
for rep in 1 2 3; do
  for rays in 128 512 1024 4096; do
    run "rolled fold loop (before)" $rays NERFDS_LIB=$OLD
    run "16 loads in flight (after)" $rays X=1
  done
done | tee $OUT/fold_ab.txt
