#!/bin/bash
# authoring container, after tools/r5_final.sh came back: the summaries of gpurun_out/ that are committed under profiles/r5_*
cd "$(dirname "$0")/.."
for k in bf16 bf16x3; do
  P=gpurun_out/profile_r5_$k
  cp $P/summary.txt profiles/r5_${k}_rocprof_summary.txt
  cp $P/hbm_traffic.json profiles/r5_${k}_hbm_traffic.json
  cp $P/bench_under_rocprof.json profiles/r5_${k}_bench_under_rocprof.json
done
cp gpurun_out/profile_r5_bf16x3_fine/summary.txt profiles/r5_bf16x3_fine_rocprof_summary.txt
cp gpurun_out/profile_r5_bf16x3_fine/bench_under_rocprof.json profiles/r5_bf16x3_fine_bench_under_rocprof.json
cp gpurun_out/prof_r5_train/summary.txt profiles/r5_train_rocprof_summary.txt
cp gpurun_out/prof_r5_train/bench_under_rocprof.json profiles/r5_train_bench_under_rocprof.json
python - <<'PY'
import json
t = json.load(open('gpurun_out/prof_r5_train_traffic/traffic.json'))
t['measured_on'] = 'round 5 (final build), separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --train --steps 3 --warmup 1 --no-cpu-baseline` (tools/prof_train_traffic.sh), FETCH_SIZE doubled per MI355X_MICROARCH.md'
json.dump(t, open('profiles/r5_train_hbm_traffic.json', 'w'), indent=1)
PY
cp gpurun_out/timeline_r5.txt profiles/r5_train_timeline.txt
cp gpurun_out/prof_r5_objective/summary.txt profiles/r5_train_full_objective_rocprof_summary.txt
cp gpurun_out/objective_timeline_r5.txt profiles/r5_train_full_objective_timeline.txt
mkdir -p profiles/r5_bench_lines; cp gpurun_out/final_r5/*.json profiles/r5_bench_lines/; cp gpurun_out/r5_final/objective_time.txt profiles/r5_bench_lines/
ls profiles | grep r5_
