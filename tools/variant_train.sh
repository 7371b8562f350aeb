#!/bin/bash
# Development: variant libraries that differ in the plain training step's fused forward kernel (the Makefile's k_train_fwd16.o).  usage: tools/variant_train.sh name:"-DFLAG ..." ...
#   result: nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_<name>.so (needs a finished `make`); run with NERFDS_LIB=<that file>
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -c render_kernel.hip -DNERFDS_TRAIN_HALF=1 -DNERFDS_TRAIN_PIPE=1 $f \
      -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_TRAIN_FWD -DNERFDS_NAME=train_fwd16_nerfds -Rpass-analysis=kernel-resource-usage -o build/abl/kt_$n.o 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c | sed "s/^/$n: /"
    others=$(ls build/k_*.o | grep -v "k_train_fwd16.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/kt_$n.o $others \
      build/host.o build/camera.o build/frame.o build/train_k.o build/train_g.o build/train.o ) &
done
wait
