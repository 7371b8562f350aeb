#!/bin/bash
# round 4, GPU call 2: pinned-order pipelined epilogue of the split-bf16 kernel (main) against the same build without it (nopin); every GPU test
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_2; mkdir -p $O
python tools/ab.py bf16x3 4 main nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_nopin.so > $O/ab_x3_pin.txt 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/gputests.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/ab_x3_pin.txt $O/gputests.log; tail -c 1500 $O/bench_default.json
