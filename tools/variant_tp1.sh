#!/bin/bash
# Development: library variant whose two-N-tile kernels take their tiles one at a time (graphs.h NERFDS_NT2_TILE_PAIR=1, host packer + kernels).  usage: tools/variant_tp1.sh <name> "<extra kernel flags>"
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
n=$1; xf=$2
H="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value"
mkdir -p build/abl/$n
$H -x hip -DNERFDS_NT2_TILE_PAIR=1 -c nerfds_host.cpp -o build/abl/$n/host.o &
for k in nerfds_bf16:GraphNerfDS:P_BF16 nerfds_f16:GraphNerfDS:P_F16 hyper_bf16:GraphHyperNeRF:P_BF16 hyper_f16:GraphHyperNeRF:P_F16; do
  nm=${k%%:*}; r=${k#*:}; g=${r%%:*}; p=${r#*:}
  ( $H -mllvm -amdgpu-mfma-vgpr-form=1 -c render_kernel.hip -fno-slp-vectorize -DNERFDS_NT=2 -DNERFDS_ASM_EPILOGUE=0 -DNERFDS_TILE_PAIR=1 $xf -DNERFDS_GRAPH=$g -DNERFDS_PREC=$p -DNERFDS_NAME=$nm -Rpass-analysis=kernel-resource-usage -o build/abl/$n/k_$nm.o 2>&1 | grep -E "error|VGPRs Spill" | sort | uniq -c | sed "s/^/$n $nm: /" ) &
done
wait
others=$(ls build/k_*.o | grep -v -E "k_(nerfds|hyper)_(bf16|f16)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so build/abl/$n/k_*.o $others build/abl/$n/host.o build/camera.o build/frame.o build/train_k.o build/train_g.o build/train.o
ls -la ../nerfds_amd/_lib/abl/libnerfds_hip_$n.so
