#!/bin/bash
# Development: variants of the bf16x3 nerf_ds kernel.  usage: tools/variant_x3.sh name:"-DFLAG"
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -c render_kernel.hip $f \
      -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16X3 -DNERFDS_NAME=nerfds_bf16x3 -Rpass-analysis=kernel-resource-usage -o build/abl/kx_$n.o 2>&1 | grep -E "error|VGPRs Spill" | sed "s/^/$n: /"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/kx_$n.o \
      build/k_nerfds_bf16.o build/k_nerfds_f32.o build/k_static_bf16.o build/k_static_bf16x3.o build/k_static_f32.o build/k_hyper_bf16.o build/k_hyper_bf16x3.o build/k_hyper_f32.o build/host.o build/camera.o build/frame.o build/train_k.o build/train.o -L/opt/rocm/lib -lrocblas -Wl,-rpath,/opt/rocm/lib ) &
done
wait
