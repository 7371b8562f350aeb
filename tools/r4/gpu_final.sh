#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_final; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gputests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/final_bench.sh r4 > $O/final_bench.log 2>&1
cat $O/gputests.log $O/smoke.log; tail -20 $O/final_bench.log | cut -c1-330
