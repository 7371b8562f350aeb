#!/bin/bash
# End of round 6 (GPU box, one call): the full GPU suite, smoke(), every bench line, and the profiles committed under profiles/r6_*:
#   rocprofv3 kernel trace + PMC passes of the default bench line (bf16) and of the parity path (split bf16), of the training step (rgb objective)
#   and of the full configs/nerf_ds.gin objective (+ its one-stream timeline), HBM traffic of the training step.
# usage: gpurun -- 'bash tools/r6_final.sh'   -> gpurun_out/r6_final/...; copy_r6_profiles.sh (below) moves the summaries into profiles/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_final; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/gputests.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 1800 bash tools/final_bench.sh r6 > $O/final_bench.log 2>&1
bash tools/prof_bench.sh r6_bf16 > /dev/null 2>&1
bash tools/prof_bench.sh r6_bf16x3 --precision bf16x3 > /dev/null 2>&1
bash tools/prof_bench.sh r6_bf16x3_fine --precision bf16x3_fine > /dev/null 2>&1
bash tools/prof_bench.sh r6_f16x3 --precision f16x3 > /dev/null 2>&1
for k in bf16 bf16x3 f16x3; do
  P=gpurun_out/profile_r6_$k
  python tools/traffic_from_summary.py $P/summary.txt "render_rays_kernel<GraphNerfDS, $k>" profiles/r6_${k}_rocprof_summary.txt "round 6 (final build)" > $P/hbm_traffic.json
done
timeout 1000 bash tools/prof_train.sh r6_train > /dev/null 2>&1
timeout 2000 bash tools/prof_train_traffic.sh r6_train_traffic > /dev/null 2>&1
timeout 1000 bash tools/train_timeline.sh r6 > /dev/null 2>&1
TAG=prof_r6_objective timeout 1000 bash tools/prof_objective.sh > /dev/null 2>&1
timeout 1000 bash tools/objective_timeline.sh r6 > /dev/null 2>&1
python tools/objective_time.py > $O/objective_time.txt 2>&1
python tools/parity_sweep.py > $O/parity_sweep.jsonl 2>/dev/null
# the shape the reference trains at (configs/nerf_ds.gin:4 batch_size = 512)
python bench.py --train --train-rays 512 --no-cpu-baseline > gpurun_out/final_r6/bench_train_reference_batch_512.json 2>/dev/null
cat $O/gputests.log $O/smoke.log $O/objective_time.txt; tail -20 $O/final_bench.log | cut -c1-330
head -12 gpurun_out/prof_r6_train/summary.txt | cut -c1-170; head -6 gpurun_out/prof_r6_train_traffic/traffic.json
timeout 600 bash tools/power_probe_r6.sh > $O/power_probe.txt 2>&1
