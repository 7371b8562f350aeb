#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
A=nerf-ds_amd/nerfds_amd/_lib/abl
{
python tools/dump_render.py /tmp/main.npz 2051 bf16x3 2>&1 | tail -1
for v in xi8 xi4 xr8; do NERFDS_LIB=$A/libnerfds_hip_$v.so python tools/dump_render.py /tmp/$v.npz 2051 bf16x3 2>&1 | tail -1; echo "cmp $v"; python tools/cmp_npz.py /tmp/$v.npz /tmp/main.npz | tail -3; done
python tools/ab.py bf16x3 3 main $A/libnerfds_hip_xi8.so $A/libnerfds_hip_xi4.so $A/libnerfds_hip_xr8.so
} > gpurun_out/r3s1/ablate4.log 2>&1
cat gpurun_out/r3s1/ablate4.log
