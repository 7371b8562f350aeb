#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
 timeout 900 python bench.py --steps 5 --warmup 2
 timeout 600 python bench.py --strong --steps 5 --warmup 2 --no-cpu-baseline
 timeout 600 python bench.py --train --steps 5 --warmup 2
 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15
} > gpurun_out/bench_r2_try.log 2>&1
cat gpurun_out/bench_r2_try.log
