#!/bin/bash
# GPU box: power / clock of the card while the bench's render kernel runs (evidence for the DVFS argument in DESIGN.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for prec in bf16 bf16x3 f16; do python bench.py --steps ${STEPS:-400} --warmup 2 --no-cpu-baseline --no-other-paths --precision $prec > gpurun_out/power_bench_$prec.json 2>/dev/null &
  pid=$!; sleep 22
  for i in 1 2 3 4 5; do echo "== $prec sample $i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor junction" ; sleep 0.5; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null; done
  echo "== idle"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; rocm-smi --showmaxpower 2>/dev/null | grep -i power ) > gpurun_out/power_probe.log 2>&1
grep -E "==|Power|sclk" gpurun_out/power_probe.log | paste - - - | head -40
