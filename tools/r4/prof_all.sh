#!/bin/bash
# round 4 profiles (GPU box): rocprofv3 kernel trace + PMC passes of the default bench line (bf16) and of the parity path (split bf16), default bench line
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r4_bf16 > /dev/null 2>&1
bash tools/prof_bench.sh r4_bf16x3 --precision bf16x3 > /dev/null 2>&1
for k in bf16 bf16x3; do
  O=gpurun_out/profile_r4_$k
  python tools/traffic_from_summary.py $O/summary.txt "render_rays_kernel<GraphNerfDS, $k>" profiles/r4_${k}_rocprof_summary.txt "round 4 (final build)" > $O/hbm_traffic.json
done
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
for k in bf16 bf16x3; do echo "== $k"; grep -E "pmc |avg_ms" gpurun_out/profile_r4_$k/summary.txt | grep -v Fill | head -40; cat gpurun_out/profile_r4_$k/hbm_traffic.json; done
tail -c 3000 gpurun_out/r4_bench_default.json
