#!/bin/bash
# round 4, GPU call 1: the refactored build through every GPU test, then an interleaved A/B of the split-bf16 kernel against the
# epilogue-twice measurement build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_1; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/gputests.log 2>&1
python tools/ab.py bf16x3 3 main nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_epi2.so > $O/ab_epi2.txt 2>&1
python tools/ab.py bf16 2 main > $O/ab_bf16.txt 2>&1
cat $O/gputests.log $O/ab_epi2.txt $O/ab_bf16.txt
