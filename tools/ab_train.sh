#!/bin/bash
# Development: bench.py --train under variant libraries (tools/variant.sh <name>:k_train_fwd16,k_train_bwd16:"<flags>").  usage: tools/ab_train.sh main nt same ...
for l in "$@"; do
  if [ $l = main ]; then unset NERFDS_LIB; else export NERFDS_LIB=$PWD/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_$l.so; fi
  echo -n "$l: "; python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_last'])"
done
