#!/bin/bash
# GPU box: the step's packing as ONE launch (k_pack_batch) against one launch per stream (the library built from the commit before), interleaved; trainer tests first
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_packold.so
run() { local label=$1 rays=$2; shift 2
  env "$@" python bench.py --train --train-rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label rays $rays: %.3f ms, full objective %.3f' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step', float('nan'))))"
}
for rep in 1 2 3; do
  for rays in 128 512 1024 4096; do
    run "one launch per stream (before)" $rays NERFDS_LIB=$OLD
    run "one launch for all (after)" $rays X=1
  done
done | tee $OUT/pack_batch_ab.txt
