#!/bin/bash
# Training-step evidence (GPU box): bench lines (MFMA mode, rocBLAS mode, full objective), kernel trace, dense-layer microbenchmark.
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_evidence
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_train.py 4096 10 > $OUT/bench_mfma.json 2>$OUT/err.log
NERFDS_TRAIN_GEMM=rocblas timeout 300 python tools/bench_train.py 4096 10 > $OUT/bench_rocblas.json 2>>$OUT/err.log
timeout 300 python tools/bench_train.py 4096 5 full > $OUT/bench_full.json 2>>$OUT/err.log
NERFDS_TRAIN_GEMM=rocblas timeout 300 python tools/bench_train.py 4096 5 full > $OUT/bench_full_rocblas.json 2>>$OUT/err.log
timeout 300 nerf-ds_amd/nerfds_amd/_lib/train_gemm_check > $OUT/dense_microbench.txt 2>>$OUT/err.log
bash tools/prof_train.sh evidence > $OUT/rocprof_summary.txt 2>>$OUT/err.log
cat $OUT/bench_*.json; tail -8 $OUT/dense_microbench.txt; head -5 $OUT/rocprof_summary.txt
