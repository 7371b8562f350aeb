#!/bin/bash
# GPU box: HBM bytes of the whole training step from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, summed over
# every kernel of `bench.py --train`), per step.  Usage: tools/prof_train_traffic.sh <tag>
set -u
TAG=${1:-train_traffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=3; WARM=1
CMD="python $GRAFT_REPO_ROOT/bench.py --train --steps $STEPS --warmup $WARM --no-cpu-baseline --no-full-objective --no-option-legs ${TRAIN_ARGS:-}"
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
python - $OUT $((STEPS + WARM)) <<'PY' | tee $OUT/traffic.json
import glob, json, os, sqlite3, sys
root, steps = sys.argv[1], int(sys.argv[2])
tot, per = {}, {}
adam = []
for db in glob.glob(os.path.join(root, '**', '*_results.db'), recursive=True):
  cur = sqlite3.connect(db).cursor()
  for k, c, s, n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    if 'k_adam(' in k or k.startswith('nerfds_train::k_adam') or k.split('(')[0].endswith('k_adam'):
      adam.append(n)        # ONE k_adam launch per step: the step count is read off the trace, not assumed (round 5's file divided 9 steps by 4)
    if 'nerfds' not in k:
      continue
    tot[c] = tot.get(c, 0.0) + s
    per.setdefault(k.split('(')[0][:70], {})[c] = (s, n)
# MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count kilobytes; FETCH_SIZE under-reports by 2x on gfx950
if adam:
  assert len(set(adam)) == 1, adam
  steps = adam[0]
fetch = 2.0 * tot.get('FETCH_SIZE', 0.0) * 1024 / steps
write = tot.get('WRITE_SIZE', 0.0) * 1024 / steps
top = sorted(per.items(), key=lambda kv: -(2 * kv[1].get('FETCH_SIZE', (0, 0))[0] + kv[1].get('WRITE_SIZE', (0, 0))[0]))[:8]
print(json.dumps({'steps_counted': steps, 'fetch_bytes_per_step': fetch, 'write_bytes_per_step': write, 'hbm_bytes_per_step': fetch + write,
                  'top_kernels_bytes_per_step': {k: {'fetch': 2 * v.get('FETCH_SIZE', (0, 0))[0] * 1024 / steps, 'write': v.get('WRITE_SIZE', (0, 0))[0] * 1024 / steps,
                                                     'launches_per_step': v.get('FETCH_SIZE', (0, 0))[1] / steps} for k, v in top}}, indent=1))
PY
rm -rf $OUT/*/*.db $OUT/*/*/*.db
