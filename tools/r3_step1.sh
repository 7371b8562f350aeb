#!/bin/bash
# round-3 step 1: new kernel vs the round-2 kernel: A/B time, smoke, GPU tests, diagnostic matrix
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
R2=nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_r2.so
for prec in f32; do for nc in 8 64; do for R in 300; do
  echo "== $prec nc=$nc R=$R"
  timeout 120 python tools/dump_render.py /tmp/x.npz $R $prec $nc 2>&1 | grep -E "saved|fault|Error|error" | head -3
done; done; done > gpurun_out/r3s1/matrix.log 2>&1
python -c "
import __graft_entry__ as g
g.smoke()
" > gpurun_out/r3s1/smoke.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3s1/gputests_all.log
{
python tools/ab.py bf16 3 main $R2
python tools/ab.py bf16x3 2 main $R2
} > gpurun_out/r3s1/ab.log 2>&1
cat gpurun_out/r3s1/matrix.log gpurun_out/r3s1/smoke.log gpurun_out/r3s1/gputests_all.log gpurun_out/r3s1/ab.log
