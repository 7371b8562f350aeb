#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
for prec in bf16 f32 bf16x3; do for nc in 8 16 32 64; do for R in 8 300; do
  echo "== $prec nc=$nc R=$R"
  timeout 120 python tools/dump_render.py /tmp/x.npz $R $prec $nc 2>&1 | grep -E "saved|fault|Error|error" | head -3
done; done; done > gpurun_out/r3s1/matrix.log 2>&1
cat gpurun_out/r3s1/matrix.log
