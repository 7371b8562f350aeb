#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-paths"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq1 -o p -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT -d $OUT/pmc_misc -o p -- $CMD > $OUT/pmc_misc.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db
grep -E "pmc |dispatch dur" $OUT/summary.txt | grep -v Fill | head -40
