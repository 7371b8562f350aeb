#!/bin/bash
# TCC misses of the split-bf16 kernel: shipped build against the one-NerfMLP-stream measurement build (tools/r4/gpu19.sh has the timing)
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_one.so
for v in main one; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_r4_l2_$v; mkdir -p $OUT
  if [ $v = one ]; then export NERFDS_LIB=$L; else unset NERFDS_LIB; fi
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-paths --precision bf16x3"
  timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
  rm -rf $OUT/*/*.db
  echo "== $v"; grep pmc $OUT/summary.txt | cut -c1-110
done
