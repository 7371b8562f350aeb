#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s1
python -c "
import __graft_entry__ as g
try:
  g.smoke()
except Exception as e:
  print('SMOKE FAILED', repr(e)[:300])
" > gpurun_out/r3s1/smoke.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3s1/gputests_all.log
cat gpurun_out/r3s1/smoke.log gpurun_out/r3s1/gputests_all.log
