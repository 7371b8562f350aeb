#!/bin/bash
# Static ISA report of one fused-kernel variant: tools/isa_report.sh <name> "<-D flags>" [kernel-name substring] [source file]
#   e.g. tools/isa_report.sh x3 "-DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16X3 -DNERFDS_NAME=nerfds_bf16x3 -fno-slp-vectorize"
# Writes /tmp/isa/<name>.s and prints tools/isa_stats.py's JSON for the non-WIDE instantiation.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
n=$1; f=$2; pat=${3:-Lb0}; src=${4:-render_kernel.hip}
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$ROOT/include -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -S \
  $ROOT/nerf-ds_amd/csrc/$src $f -o /tmp/isa/$n.s 2>&1 | grep -v "hip-link" || true
python $ROOT/tools/isa_stats.py /tmp/isa/$n.s $pat --top ${TOP:-16}
