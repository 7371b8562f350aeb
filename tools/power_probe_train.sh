#!/bin/bash
# GPU box: power / clock of the card while bench.py --train is in its timed loop (is the training step at the power limit?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for mode in 1 0; do NERFDS_TRAIN_FUSED_FWD=$mode python bench.py --train --steps ${STEPS:-900} --warmup 3 --no-cpu-baseline > gpurun_out/power_train_$mode.json 2>/dev/null &
  pid=$!; sleep 20
  for i in 1 2 3 4 5 6; do echo "== fused_fwd=$mode sample $i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" ; sleep 0.4; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null; done
  echo "== idle"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" ) > gpurun_out/power_probe_train.log 2>&1
grep -E "==|Power|sclk" gpurun_out/power_probe_train.log | paste - - - | head -20
