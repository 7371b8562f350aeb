#!/bin/bash
# end of round 4: full GPU suite, smoke, every bench line, profiles of the training step and of the two render kernels (each step under its own timeout)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_final2; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gputests.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 1800 bash tools/final_bench.sh r4b > $O/final_bench.log 2>&1
timeout 1000 bash tools/prof_train.sh r4_train > /dev/null 2>&1
timeout 2000 bash tools/prof_train_traffic.sh r4_train_traffic > /dev/null 2>&1
timeout 1000 bash tools/train_timeline.sh r4 > /dev/null 2>&1
cat $O/gputests.log $O/smoke.log; tail -20 $O/final_bench.log | cut -c1-330; head -12 gpurun_out/prof_r4_train/summary.txt | cut -c1-170; head -6 gpurun_out/prof_r4_train_traffic/traffic.json
