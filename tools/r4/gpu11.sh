#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_11; mkdir -p $O; X=$PWD/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_x6.so
for rep in 1 2; do
echo -n "main: "; python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_last'])"
echo -n "x6:   "; NERFDS_LIB=$X python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_last'])"
done > $O/ab_x6.txt 2>&1
( NERFDS_LIB=$X timeout 1500 python -m pytest tests/test_training.py tests/test_golden.py -m gpu -q -s 2>&1 | grep -E "^E  |passed|failed|FAILED|worst leaves" | head -40 ) > $O/x6_tests.log 2>&1
cat $O/ab_x6.txt $O/x6_tests.log
