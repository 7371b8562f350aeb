#!/bin/bash
# GPU box: SQ counters of the trainer's kernels (one rocprofv3 --pmc pass, kernel-serialised): matrix-pipe busy fraction, instruction mix and waits per kernel.
# Usage: tools/prof_train_sq.sh [tag]   -> gpurun_out/train_sq_<tag>/summary.txt
set -u
TAG=${1:-r6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/${BENCH_SCRIPT:-bench.py --train} --steps 3 --warmup 1 --no-cpu-baseline --no-full-objective --no-option-legs ${TRAIN_ARGS:-}"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS -d $OUT/pmc -o p -- $CMD > $OUT/pmc.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
python - <<PY > $OUT/summary.txt
import glob, sqlite3, collections
res = collections.defaultdict(dict)
for db in glob.glob('$OUT/pmc*/**/*_results.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    for k, c, s, n, d in cur.execute("select kernel_name, counter_name, sum(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
        res[k][c] = (s / n, n, d)
rows = []
for k, v in res.items():
    if 'SQ_INSTS_MFMA' not in v or 'GRBM_GUI_ACTIVE' not in v: continue
    g = lambda c: v.get(c, (0, 0, 0))[0]
    n, d = v['SQ_INSTS_MFMA'][1], v['SQ_INSTS_MFMA'][2]
    cyc = g('GRBM_GUI_ACTIVE')
    # busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE): the normalisation tools/prof_summary.py uses for the render kernels (71 % for split bf16)
    rows.append((n * d, k, n, d, g('SQ_VALU_MFMA_BUSY_CYCLES') / (128.0 * cyc) if cyc else 0, g('SQ_INSTS_MFMA'), g('SQ_INSTS_VALU'), g('SQ_INSTS_VMEM'), g('SQ_INSTS_LDS'),
                 g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else 0, g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else 0))
print('kernel | launches | avg us (serialised, under PMC) | MFMA pipe busy | MFMA insts | VALU / MFMA | VMEM / MFMA | LDS / MFMA | wait-inst / wave cycles | active-inst / wave cycles')
for _, k, n, d, busy, mf, va, vm, ld, wt, ac in sorted(rows, reverse=True)[:16]:
    print('%s | %d | %.1f | %.3f | %.3g | %.2f | %.3f | %.2f | %.2f | %.2f' % (k[:88], n, d / 1e3, busy, mf, va / mf if mf else 0, vm / mf if mf else 0, ld / mf if mf else 0, wt, ac))
PY
rm -rf $OUT/pmc $OUT/pmc2
cat $OUT/summary.txt
