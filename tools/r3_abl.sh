#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
A=nerf-ds_amd/nerfds_amd/_lib/abl
{
NERFDS_LIB=$PWD/$A/libnerfds_hip_rk.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_camera.py tests/test_render_image_gpu.py -q -m gpu -k "not rccl and not nccl and not two_ranks" 2>&1 | tail -3
python tools/ab.py bf16 3 main $A/libnerfds_hip_dpp.so $A/libnerfds_hip_rk.so
python tools/ab.py bf16x3 2 main $A/libnerfds_hip_rkx.so
} > gpurun_out/r3s1/ablate6.log 2>&1
cat gpurun_out/r3s1/ablate6.log
