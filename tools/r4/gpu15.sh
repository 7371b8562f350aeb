#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_15; mkdir -p $O
( timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log 2>&1
for i in 1 2; do
python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('multi', d['ms_per_step'])" >> $O/bench.log 2>&1
NERFDS_WGRAD_MULTI=0 python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', d['ms_per_step'])" >> $O/bench.log 2>&1
done
for n in 1 2 4; do NERFDS_TRAIN_SIDE_STREAMS=$n python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('multi side $n', d['ms_per_step'])"; done >> $O/bench.log
bash tools/train_timeline.sh r4c > /dev/null 2>&1
cat $O/tests.log | cut -c1-250; cat $O/bench.log
