#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_8; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/gputests.log 2>&1
bash tools/r4/prof_all.sh > $O/prof_all.log 2>&1
cat $O/gputests.log; tail -60 $O/prof_all.log | cut -c1-300
