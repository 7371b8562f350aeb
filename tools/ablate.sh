#!/bin/bash
# Development: build the bf16 nerf_ds kernel with an ablation mask into variant libraries (run from the repo root, CPU box).
set -e
cd "$(dirname "$0")/../nerf-ds_amd/csrc"
mkdir -p build/abl ../nerfds_amd/_lib/abl
for m in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -c render_kernel.hip \
      -DNERFDS_ABLATE=$m -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16 -DNERFDS_NAME=nerfds_bf16 -o build/abl/k_$m.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$m.so build/abl/k_$m.o \
      build/k_nerfds_bf16x3.o build/k_nerfds_f32.o build/k_static_bf16.o build/k_static_bf16x3.o build/k_static_f32.o build/host.o ) &
done
wait
ls -la ../nerfds_amd/_lib/abl/
