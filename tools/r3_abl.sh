#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
A=nerf-ds_amd/nerfds_amd/_lib/abl
{
NERFDS_LIB=$PWD/$A/libnerfds_hip_cry.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_render_image_gpu.py -q -s -m gpu -k "not rccl and not nccl and not two_ranks and not training" 2>&1 | grep -E "full-frame|oracle sub|passed|failed|^FAILED|bf16x3" | tail -30
python tools/ab.py bf16x3 3 main $A/libnerfds_hip_hws.so $A/libnerfds_hip_cry.so
} > gpurun_out/r3s1/ablate8.log 2>&1
cat gpurun_out/r3s1/ablate8.log
