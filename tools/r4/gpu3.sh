#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_3; mkdir -p $O; A=nerf-ds_amd/nerfds_amd/_lib/abl
python tools/ab.py bf16x3 3 main $A/libnerfds_hip_nolgkm.so $A/libnerfds_hip_pin2.so $A/libnerfds_hip_pin3.so $A/libnerfds_hip_nopin.so > $O/ab_x3.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_render_image_gpu.py -m gpu -q 2>&1 | tail -5 ) > $O/tests_main.log 2>&1
( NERFDS_LIB=$PWD/$A/libnerfds_hip_pin2.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -5 ) > $O/tests_pin2.log 2>&1
cat $O/ab_x3.txt $O/tests_main.log $O/tests_pin2.log
