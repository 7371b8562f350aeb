#!/bin/bash
# GPU box: profile `python bench.py` with rocprofv3 (kernel trace + stats, then separate PMC passes) and write a
# text summary next to the databases.  Usage: tools/prof_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-paths $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq1 -o p -- $CMD > $OUT/pmc_sq1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $OUT/pmc_misc -o p -- $CMD > $OUT/pmc_misc.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $OUT/*/*.db    # keep the text, not the 5 MB databases
cat $OUT/summary.txt
