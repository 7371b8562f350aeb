#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s3
timeout 600 python bench.py --graph static > gpurun_out/r3s3/bench_static.json 2> gpurun_out/r3s3/bench_static.err
bash tools/prof_bench.sh r3 > gpurun_out/r3s3/prof_bf16.log 2>&1
bash tools/prof_bench.sh r3_bf16x3 --precision bf16x3 > gpurun_out/r3s3/prof_bf16x3.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r3s3/gputests.log
tail -3 gpurun_out/r3s3/bench_static.err; head -c 1500 gpurun_out/r3s3/bench_static.json; tail -5 gpurun_out/r3s3/gputests.log
