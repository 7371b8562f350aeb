#!/bin/bash
# Development: emit device asm for every (graph, precision) kernel and run tools/isa_lint.py on it.
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/asm
for g in nerfds:GraphNerfDS static:GraphStatic hyper:GraphHyperNeRF; do for p in bf16:P_BF16 bf16x3:P_BF16X3 f32:P_F32; do
  n=${g%%:*}_${p%%:*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value --cuda-device-only -S render_kernel.hip \
     -DNERFDS_GRAPH=${g#*:} -DNERFDS_PREC=${p#*:} -DNERFDS_NAME=$n -o build/asm/$n.s 2>/dev/null &
done; done
wait
python3 ../../tools/isa_lint.py build/asm/*.s | grep -v "^$"
