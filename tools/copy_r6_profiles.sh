#!/bin/bash
# authoring container, after tools/r6_final.sh came back: the summaries of gpurun_out/ that are committed under profiles/r6_*
cd "$(dirname "$0")/.."
for k in bf16 bf16x3 f16x3; do
  P=gpurun_out/profile_r6_$k
  cp $P/summary.txt profiles/r6_${k}_rocprof_summary.txt
  cp $P/hbm_traffic.json profiles/r6_${k}_hbm_traffic.json
  cp $P/bench_under_rocprof.json profiles/r6_${k}_bench_under_rocprof.json
done
cp gpurun_out/profile_r6_bf16x3_fine/summary.txt profiles/r6_bf16x3_fine_rocprof_summary.txt
cp gpurun_out/profile_r6_bf16x3_fine/bench_under_rocprof.json profiles/r6_bf16x3_fine_bench_under_rocprof.json
cp gpurun_out/prof_r6_train/summary.txt profiles/r6_train_rocprof_summary.txt
cp gpurun_out/prof_r6_train/bench_under_rocprof.json profiles/r6_train_bench_under_rocprof.json
python - <<'PY'
import json
t = json.load(open('gpurun_out/prof_r6_train_traffic/traffic.json'))
t['measured_on'] = 'round 6 (final build), separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --train --steps 3 --warmup 1 --no-cpu-baseline --no-full-objective --no-option-legs` (tools/prof_train_traffic.sh; steps counted from the k_adam launches), FETCH_SIZE doubled per MI355X_MICROARCH.md'
json.dump(t, open('profiles/r6_train_hbm_traffic.json', 'w'), indent=1)
PY
cp gpurun_out/timeline_r6.txt profiles/r6_train_timeline.txt
cp gpurun_out/prof_r6_objective/summary.txt profiles/r6_train_full_objective_rocprof_summary.txt
cp gpurun_out/objective_timeline_r6.txt profiles/r6_train_full_objective_timeline.txt
cp gpurun_out/r6_final/parity_sweep.jsonl profiles/r6_parity_sweep.jsonl
mkdir -p profiles/r6_bench_lines; cp gpurun_out/final_r6/*.json profiles/r6_bench_lines/; cp gpurun_out/r6_final/objective_time.txt profiles/r6_bench_lines/
ls profiles | grep r6_
