#!/bin/bash
# Profiling recipe (GPU box): kernel trace + separate PMC passes. Usage: tools/prof.sh <tag> <rays> <prec>
set -u
TAG=${1:-p}; R=${2:-16384}; PREC=${3:-bf16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/quick_time.py $R $PREC"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $OUT/pmc5 -o p -- $CMD > $OUT/pmc5.log 2>&1
find $OUT -name "*.csv" | head -50
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -5 $f; done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/pmc*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'render_rays' not in r.get('Kernel_Name', ''): continue
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        for k, (v, n) in agg.items(): print(d.split('/')[-1], k, 'sum', v, 'dispatches', n, 'per-dispatch', v / max(n, 1))
PY
