#!/bin/bash
# GPU box: socket power / shader clock while the round-6 kernels run (rocm-smi sampled 5 x, 20 s into the timed loop): the render kernel in the three
# arithmetics of the bench line and the training step at BASELINE config 4's batch and at the reference's own (512).  Output: gpurun_out/power_probe_r6.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
probe() {  # label, command...
  local label=$1; shift
  "$@" > /dev/null 2>&1 &
  local pid=$!; sleep ${WARM:-20}
  for i in 1 2 3 4 5; do echo "== $label sample $i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 0.5; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
}
( for prec in bf16 bf16x3 f16x3 bf16x3_fine; do probe $prec python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-other-paths --precision $prec; done
  probe train_4096 python bench.py --train --steps 3000 --warmup 2 --no-cpu-baseline --no-full-objective --no-option-legs
  probe train_512 python bench.py --train --train-rays 512 --steps 40000 --warmup 2 --no-cpu-baseline --no-full-objective --no-option-legs
  echo "== idle"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; rocm-smi --showmaxpower 2>/dev/null | grep -i power ) > gpurun_out/power_probe_r6.log 2>&1
grep -E "==|Power|sclk" gpurun_out/power_probe_r6.log | paste - - - | cut -c1-230 | head -40
