#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_5; mkdir -p $O; A=nerf-ds_amd/nerfds_amd/_lib/abl
( timeout 600 python -m pytest "tests/test_gpu_parity.py" -m gpu -q -k "bf16 or f16 or determin" 2>&1 | grep -E "^E|assert|passed|failed|bf16 |f16 " | head -60 ) > $O/t_main.log 2>&1
( NERFDS_LIB=$PWD/$A/libnerfds_hip_j2.so timeout 600 python -m pytest "tests/test_gpu_parity.py" -m gpu -q -k "bf16 or f16 or determin" 2>&1 | grep -E "^E|passed|failed" | head -30 ) > $O/t_j2.log 2>&1
( NERFDS_LIB=$PWD/$A/libnerfds_hip_old.so timeout 600 python -m pytest "tests/test_gpu_parity.py" -m gpu -q -k "tiny" 2>&1 | grep -E "^E|passed|failed|bf16 " | head -30 ) > $O/t_old.log 2>&1
echo MAIN; cat $O/t_main.log; echo J2; cat $O/t_j2.log; echo OLD; cat $O/t_old.log
