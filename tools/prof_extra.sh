#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_extra
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/quick_time.py 65536 bf16"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT | grep "pmc "
tail -3 $OUT/p1.log | head -2
