#!/bin/bash
# GPU box: the bench lines quoted in DESIGN section 6 / README, written to gpurun_out/final_<tag>/ (copied to profiles/ by hand)
TAG=${1:-r2}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --precision bf16x3 --steps 3 --warmup 1 --no-other-paths --no-cpu-baseline > $OUT/bench_bf16x3.json 2>/dev/null
python bench.py --precision f32 --steps 2 --warmup 1 --no-other-paths --no-cpu-baseline > $OUT/bench_f32.json 2>/dev/null
python bench.py --strong --no-cpu-baseline > $OUT/bench_strong.json 2>/dev/null
python bench.py --graph hypernerf --samples 128 --steps 5 --warmup 2 --no-other-paths --no-cpu-baseline > $OUT/bench_config5_hypernerf_128.json 2>/dev/null
python bench.py --train > $OUT/bench_train.json 2>/dev/null
for f in $OUT/*.json; do echo "== $f"; tail -1 $f | cut -c1-400; done
