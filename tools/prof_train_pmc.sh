#!/bin/bash
# GPU box: HBM traffic of the trainer's kernels (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE as
# MI355X_MICROARCH.md prescribes).  Usage: tools/prof_train_pmc.sh   -> gpurun_out/train_pmc/summary.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_train.py 4096 2"
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python - <<PY > $OUT/summary.txt
import glob, sqlite3, collections
res = collections.defaultdict(dict)
for tag in ('fetch', 'write'):
    for db in glob.glob('$OUT/%s/**/*_results.db' % tag, recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, c, s, n, d in cur.execute("select kernel_name, counter_name, sum(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
            res[k][c] = (s / n, n, d)
rows = []
for k, v in res.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        f, n, d = v['FETCH_SIZE']; w = v['WRITE_SIZE'][0]
        rows.append((n * d, k, n, d, 2 * f * 1024, w * 1024))          # FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md); KB units
print('kernel | launches | avg us (under PMC) | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch | GB/s')
for _, k, n, d, f, w in sorted(rows, reverse=True)[:14]:
    print('%s | %d | %.1f | %.1f | %.1f | %.0f' % (k[:70], n, d / 1e3, f / 1e6, w / 1e6, (f + w) / d))
PY
rm -rf $OUT/*/*.db $OUT/fetch $OUT/write 2>/dev/null
cat $OUT/summary.txt
