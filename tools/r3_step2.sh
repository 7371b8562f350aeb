#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s2
A=nerf-ds_amd/nerfds_amd/_lib/abl
timeout 900 python -m pytest tests/test_rccl_single_gpu.py tests/test_render_image_gpu.py -m gpu -q -s -k "rccl or full_frame" 2>&1 | tail -25 > gpurun_out/r3s2/newtests.log
timeout 600 python bench.py > gpurun_out/r3s2/bench_default.json 2> gpurun_out/r3s2/bench_default.err
timeout 600 python bench.py --graph static > gpurun_out/r3s2/bench_static.json 2> gpurun_out/r3s2/bench_static.err
timeout 600 python bench.py --sweep --steps 3 > gpurun_out/r3s2/bench_sweep.json 2> gpurun_out/r3s2/bench_sweep.err
timeout 600 python bench.py --precision mixed --steps 3 --no-cpu-baseline > gpurun_out/r3s2/bench_mixed.json 2> gpurun_out/r3s2/bench_mixed.err
python tools/ab.py bf16x3 3 main $A/libnerfds_hip_xi6.so > gpurun_out/r3s2/ab_xi6.log 2>&1
cat gpurun_out/r3s2/newtests.log; tail -c 600 gpurun_out/r3s2/*.err; cat gpurun_out/r3s2/ab_xi6.log
