#!/bin/bash
# GPU box: power / clock while the render kernels of the final round-3 build run, plus the 8-wave build of the bf16 kernel, the default until the end of round 3 (NERFDS_LIB variant
# from tools/variant_k.sh 'w8:bf16:-fno-slp-vectorize') and the training step.  Output: gpurun_out/power_probe_r3.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
probe() {  # label, command...
  local label=$1; shift
  "$@" > /dev/null 2>&1 &
  local pid=$!; sleep ${WARM:-20}
  for i in 1 2 3 4 5; do echo "== $label sample $i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 0.5; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
}
( for prec in bf16 bf16x3 f16; do probe $prec python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-other-paths --precision $prec; done
  NERFDS_LIB=$PWD/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_w8.so probe bf16_8wave python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-other-paths --precision bf16
  probe train python bench.py --train --steps 2000 --warmup 2 --no-cpu-baseline
  echo "== idle"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; rocm-smi --showmaxpower 2>/dev/null | grep -i power ) > gpurun_out/power_probe_r3.log 2>&1
grep -E "==|Power|sclk" gpurun_out/power_probe_r3.log | paste - - - | cut -c1-230 | head -40
