#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_23; mkdir -p $O
C=nerf-ds_amd/nerfds_amd/_lib/train_gemm_check
for r in 40 70001 524288; do echo "== rows $r"; timeout 300 $C $r 2>&1 | grep -E "head|FAIL|!!" ; done > $O/check.log 2>&1
( timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py tests/test_train_gemm.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log 2>&1
for i in 1 2; do
timeout 300 python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['ms_per_step'], d['loss_last'])" >> $O/bench.log 2>&1
NERFDS_WGRAD_HEAD_OFF=1 timeout 300 python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['loss_last'])" >> $O/bench.log 2>&1
done
cat $O/check.log | cut -c1-200; cat $O/tests.log | cut -c1-250; cat $O/bench.log
