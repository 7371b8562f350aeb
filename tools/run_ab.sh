#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
A=$PWD/nerf-ds_amd/nerfds_amd/_lib/abl
{
 timeout 900 python tools/ab.py bf16 3 main $A/libnerfds_hip_b_prio.so $A/libnerfds_hip_b_prio3.so $A/libnerfds_hip_b_r2.so
 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
