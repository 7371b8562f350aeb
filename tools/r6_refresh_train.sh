#!/bin/bash
# GPU box, after the side streams' CU cap (end of round 6): the full GPU suite on a fresh lease and the bench lines / profiles the trainer's launch rules change
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_refresh; mkdir -p $O
N=${1:-18}
( timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/gputests_$N.log 2>&1
tail -1 $O/gputests_$N.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --train > $O/bench_config4_train.json 2>/dev/null
python bench.py --train --train-rays 512 --no-cpu-baseline > $O/bench_train_reference_batch_512.json 2>/dev/null
TAG=prof_r6b_objective timeout 1000 bash tools/prof_objective.sh > /dev/null 2>&1
cp gpurun_out/prof_r6b_objective/summary.txt $O/train_full_objective_rocprof_summary.txt
python tools/objective_time.py > $O/objective_time.txt 2>&1
python - <<'PY'
import json
for f in ('bench_default', 'bench_config4_train', 'bench_train_reference_batch_512'):
  r = json.loads(open(f'gpurun_out/r6_refresh/{f}.json').read().strip().splitlines()[-1])
  if 'train_step' in r:
    print(f, r['value'], r['roofline']['frac'], r['value_at_tolerance'], r['roofline_frac_at_tolerance'], 'train', r['train_step']['ms_per_step'], r['train_step'].get('full_objective', {}).get('ms_per_step'), 'ref batch', r['train_step_reference_batch'])
  else:
    print(f, r['ms_per_step'], r['roofline']['frac'], r.get('full_objective', {}).get('ms_per_step'))
PY
cat $O/objective_time.txt | tail -12
