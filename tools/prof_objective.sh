#!/bin/bash
# rocprofv3 kernel stats of the 4096-ray step under the full nerf_ds.gin objective (second-order norm loss): where the 94 ms go
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-prof_objective}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one_case.py <<'PY'
import sys, runpy
sys.argv = ['x']
src = open(sys.argv[0] if False else '/root/repo/tools/objective_time.py').read().replace("cases = {", "cases = {k: v for k, v in {").replace("'rgb + elastic': dict(elastic_loss_weight=0.01, elastic_reduce_method='weight')}", "'rgb + elastic': dict(elastic_loss_weight=0.01, elastic_reduce_method='weight')}.items() if 'second-order' in k}")
import os; os.chdir('/root/repo'); exec(compile(src, 'objective_time', 'exec'))
PY
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python /tmp/one_case.py > $OUT/trace.log 2>&1
python - <<PY > $OUT/summary.txt
import glob, sqlite3
for db in glob.glob('$OUT/trace/**/*_results.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 40').fetchall()
    tot = sum(r[2] for r in cur.execute('select name,total_calls,total_duration from top_kernels').fetchall())
    print('total kernel time ms', tot / 1e3, '(13 steps)')
    for name, calls, t, avg, pct in rows:
        print(f'{name[:100]:100s} calls={calls:5d} total_ms={t/1e3:9.3f} avg_ms={avg/1e3:8.4f} pct={pct:5.1f}')
PY
rm -rf $OUT/trace; head -32 $OUT/summary.txt | cut -c1-170; tail -3 $OUT/trace.log
