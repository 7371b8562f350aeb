#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_4; mkdir -p $O; A=nerf-ds_amd/nerfds_amd/_lib/abl
python tools/ab.py bf16 4 main $A/libnerfds_hip_old.so $A/libnerfds_hip_j2.so $A/libnerfds_hip_nopin1.so > $O/ab_bf16.txt 2>&1
python tools/ab.py f16 3 main $A/libnerfds_hip_old.so > $O/ab_f16.txt 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/tests_main.log 2>&1
cat $O/ab_bf16.txt $O/ab_f16.txt $O/tests_main.log
