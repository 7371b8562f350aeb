#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_12; mkdir -p $O
C=nerf-ds_amd/nerfds_amd/_lib/train_gemm_check
for r in 40 16416 70001 786432 524288; do echo "== rows $r"; timeout 300 $C $r 2>&1 | grep -E "wgrad|FAIL|!!" ; done > $O/check.log 2>&1
NERFDS_WGRAD_TR_OFF=1 timeout 300 $C 786432 2>&1 | grep wgrad16 > $O/check_old.log
cat $O/check.log | cut -c1-200; echo OLD; cat $O/check_old.log | cut -c1-200
