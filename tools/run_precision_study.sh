#!/bin/bash
# GPU box: precision study over the main library and every mixed-plan variant under _lib/abl (libnerfds_hip_m_*.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=nerf-ds_amd/nerfds_amd/_lib
R=${1:-4096}
{
  timeout 600 python tools/precision_study.py bf16,f16,mixed,bf16x3 $R
  for v in $L/abl/libnerfds_hip_m_*.so; do
    NERFDS_LIB=$PWD/$v timeout 300 python tools/precision_study.py mixed $R
  done
} > gpurun_out/precision_study.log 2>&1
grep -E "rgb max-rel|time " gpurun_out/precision_study.log | tail -80
