#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_13; mkdir -p $O
( timeout 1500 python -m pytest tests/test_training.py tests/test_train_gemm.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -25 ) > $O/tests.log 2>&1
for i in 1 2; do
python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'])" >> $O/bench.log 2>&1
NERFDS_WGRAD_TR_OFF=1 python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old-kernel', d['ms_per_step'])" >> $O/bench.log 2>&1
done
cat $O/tests.log | cut -c1-250; cat $O/bench.log
