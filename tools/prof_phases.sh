#!/bin/bash
# GPU box: where a wave of the fused render kernel spends its cycles, from a MEASUREMENT build (-DNERFDS_PROF=1: s_memtime at the entry / exit of every
# layer chain, every field evaluation, compositing and resampling; ~2 % slower than the shipped kernel).  Build first, in the authoring container:
#   tools/variant.sh "prof:k_nerfds_bf16x3,k_nerfds_bf16,host:-DNERFDS_PROF=1"
# usage: tools/prof_phases.sh  -> gpurun_out/prof_phases.txt
L=nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_prof.so
NERFDS_LIB=$L python tools/quick_time.py 65536 bf16x3,bf16 2>&1 | grep -E "PROF|R=" | tee gpurun_out/prof_phases.txt
