#!/bin/bash
# Development: variant libraries that differ in ONE nerf_ds render kernel.  usage: tools/variant_k.sh name:prec:"-DFLAG ..." ...
#   prec = bf16|bf16x3|f32|f16|mixed ; result: nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_<name>.so (needs a finished `make`)
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
for v in "$@"; do
  n=${v%%:*}; r=${v#*:}; p=${r%%:*}; f=${r#*:}
  case $p in bf16) P=P_BF16;; bf16x3) P=P_BF16X3;; f32) P=P_F32;; f16) P=P_F16;; mixed) P="P_BF16 -DNERFDS_MIXED";; esac
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -c render_kernel.hip $f \
      -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=$P -DNERFDS_NAME=nerfds_$p -Rpass-analysis=kernel-resource-usage -o build/abl/kv_$n.o 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c | sed "s/^/$n: /"
    others=$(ls build/k_*.o | grep -v "k_nerfds_$p.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/kv_$n.o $others \
      build/host.o build/camera.o build/frame.o build/train_k.o build/train_g.o build/train.o ) &
done
wait
