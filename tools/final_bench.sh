#!/bin/bash
# GPU box: the bench lines quoted in DESIGN section 6 / README, written to gpurun_out/final_<tag>/ (copied to profiles/<tag>_bench_lines/ by hand)
TAG=${1:-r3}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --graph static > $OUT/bench_config1_static.json 2>/dev/null
python bench.py --precision f32 --steps 2 --warmup 1 --no-other-paths --no-cpu-baseline > $OUT/bench_f32.json 2>/dev/null
python bench.py --strong --no-cpu-baseline > $OUT/bench_config3_strong_n1.json 2>/dev/null
python bench.py --sweep --steps 3 > $OUT/bench_config5_sweep.json 2>/dev/null
python bench.py --samples 128 --steps 5 --warmup 2 --no-other-paths --no-cpu-baseline > $OUT/bench_nerfds_256samples.json 2>/dev/null
python bench.py --precision bf16x3_fine --no-other-paths --no-cpu-baseline > $OUT/bench_bf16x3_fine.json 2>/dev/null
python bench.py --precision f16x3 --no-other-paths --no-cpu-baseline > $OUT/bench_f16x3.json 2>/dev/null
python bench.py --train > $OUT/bench_config4_train.json 2>/dev/null
NERFDS_TRAIN_FUSED_BWD=0 python bench.py --train --no-cpu-baseline > $OUT/bench_config4_train_layerwise_backward.json 2>/dev/null
for f in $OUT/*.json; do echo "== $f"; tail -1 $f | cut -c1-300; done
