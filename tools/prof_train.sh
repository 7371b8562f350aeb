#!/bin/bash
# Kernel trace of the training step (GPU box): rocprofv3 --kernel-trace --stats of `bench.py --train`. Usage: tools/prof_train.sh <tag>
TAG=${1:-train_r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --train --steps 6 --warmup 3 --no-cpu-baseline --no-full-objective --no-option-legs ${TRAIN_ARGS:-} > $OUT/trace.log 2>&1
python - <<PY > $OUT/summary.txt
import glob, sqlite3
for db in glob.glob('$OUT/trace/**/*_results.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 60').fetchall()
    tot = sum(r[2] for r in cur.execute('select name,total_calls,total_duration from top_kernels').fetchall())
    adam = [r[1] for r in cur.execute('select name,total_calls from top_kernels').fetchall() if r[0].split('(')[0].endswith('k_adam')]
    print('total kernel time ms', tot / 1e3, f'({adam[0] if adam else "?"} steps of bench.py --train by the k_adam launch count: 3 warm-up + 6 timed, no option legs)')   # top_kernels durations are microseconds
    for name, calls, t, avg, pct in rows:
        print(f'{name[:110]:110s} calls={calls:5d} total_ms={t/1e3:9.3f} avg_ms={avg/1e3:8.4f} pct={pct:5.1f}')
PY
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $OUT/trace
cat $OUT/summary.txt
