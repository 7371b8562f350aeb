#!/bin/bash
# Development: variant libraries with another NERFDS_PREC_MIXED plan (kernel + host packer rebuilt with the flags).
# usage: tools/variant_mix.sh name:"-DNERFDS_MIX_MASK=P_F16 ..." ...   ->  nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_<name>.so
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value $f"
    /opt/rocm/bin/hipcc $FL -mllvm -amdgpu-mfma-vgpr-form=1 -c render_kernel.hip -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16 -DNERFDS_MIXED -DNERFDS_NAME=nerfds_mixed \
      -Rpass-analysis=kernel-resource-usage -o build/abl/km_$n.o 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c | sed "s/^/$n: /"
    /opt/rocm/bin/hipcc $FL -x hip -c nerfds_host.cpp -o build/abl/host_$n.o 2>&1 | grep -E "error"
    others=$(ls build/k_*.o | grep -v k_nerfds_mixed.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/km_$n.o build/abl/host_$n.o $others \
      build/camera.o build/frame.o build/train_k.o build/train_g.o build/train.o ) &
done
wait
