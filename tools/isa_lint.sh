#!/bin/bash
# Development / build check: emit device asm for every render kernel the library ships (3 graphs x 5 arithmetic modes + the training
# forward), with the Makefile's flags, and run tools/isa_lint.py on it.  Exit status 1 when a suspect copy is found.
cd "$(dirname "$0")/../nerf-ds_amd/csrc" || exit 1
mkdir -p build/asm
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -S render_kernel.hip"
jobs_n=0
for g in nerfds:GraphNerfDS static:GraphStatic hyper:GraphHyperNeRF; do for p in bf16:P_BF16 bf16x3:P_BF16X3 f32:P_F32 f16:P_F16 "mixed:P_BF16 -DNERFDS_MIXED"; do
  n=${g%%:*}_${p%%:*}
  XF=""; case $n in static_bf16|static_f16) XF="-fno-slp-vectorize";; *_bf16|*_f16) XF="-fno-slp-vectorize -DNERFDS_NT=2 -DNERFDS_ASM_EPILOGUE=0";; esac      # the Makefile's XFLAGS_<kernel>
  /opt/rocm/bin/hipcc $FL $XF $MIXFLAGS -DNERFDS_GRAPH=${g#*:} -DNERFDS_PREC=${p#*:} -DNERFDS_NAME=$n -o build/asm/$n.s 2>/dev/null &
  jobs_n=$((jobs_n + 1)); if [ $jobs_n -ge 8 ]; then wait -n; jobs_n=$((jobs_n - 1)); fi
done; done
for h in 0:"" 1:16; do hv=${h%%:*}; sfx=${h#*:}; TF="-DNERFDS_TRAIN_HALF=$hv"; [ $hv = 1 ] && TF="$TF -DNERFDS_TRAIN_PIPE=1"       # the Makefile's two builds of each
  /opt/rocm/bin/hipcc $FL $TF -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_TRAIN_FWD -DNERFDS_NAME=train_fwd${sfx}_nerfds -o build/asm/train_fwd$sfx.s 2>/dev/null &
  /opt/rocm/bin/hipcc $FL $TF -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_TRAIN_BWD -DNERFDS_NAME=train_bwd${sfx}_nerfds -o build/asm/train_bwd$sfx.s 2>/dev/null &
done
wait
python3 ../../tools/isa_lint.py build/asm/*.s | grep -v "^$"
exit ${PIPESTATUS[0]}
