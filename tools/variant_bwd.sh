#!/bin/bash
# Development: variant libraries that differ in the plain training step's fused forward AND backward kernels (the *16 builds of the Makefile:
# -DNERFDS_TRAIN_HALF=1 -DNERFDS_TRAIN_PIPE=1 unless BASE16 overrides them).  usage: tools/variant_bwd.sh name:"-DFLAG ..." ...
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
BASE16=${BASE16--DNERFDS_TRAIN_HALF=1 -DNERFDS_TRAIN_PIPE=1}
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( for k in FWD BWD; do lc=$(echo $k | tr A-Z a-z)
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -c render_kernel.hip $BASE16 $f \
        -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_TRAIN_$k -DNERFDS_NAME=train_${lc}16_nerfds -Rpass-analysis=kernel-resource-usage -o build/abl/kt${lc}_$n.o 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c | sed "s/^/$n $k: /" &
    done; wait
    others=$(ls build/k_*.o | grep -v "k_train_fwd16.o\|k_train_bwd16.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/ktfwd_$n.o build/abl/ktbwd_$n.o $others \
      build/host.o build/camera.o build/frame.o build/train_k.o build/train_g.o build/train.o ) &
done
wait
