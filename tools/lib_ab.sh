#!/bin/bash
# GPU box: the training step of the in-tree library against another build of it (NERFDS_LIB), interleaved on one box.
#   usage: tools/lib_ab.sh <other .so> <label of the other> <label of the in-tree one> <out name> ["<rays list>"]
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
OLD=$1; LA=$2; LB=$3; NAME=$4; RAYS=${5:-"128 512 1024 4096"}
run() { local label=$1 rays=$2; shift 2
  env "$@" python bench.py --train --train-rays $rays --steps 30 --warmup 5 --no-cpu-baseline --no-option-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label | rays $rays | %.3f ms | full objective %.3f' % (r['ms_per_step'], r.get('full_objective',{}).get('ms_per_step', float('nan'))))"
}
for rep in 1 2 3; do
  for rays in $RAYS; do
    run "$LA" $rays NERFDS_LIB=$OLD
    run "$LB" $rays X=1
  done
done | tee $OUT/$NAME.txt
python - $OUT/$NAME.txt <<'PY'
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
  p = [x.strip() for x in l.split('|')]
  if len(p) == 4: d[(p[0], int(p[1].split()[1]))].append((float(p[2].split()[0]), float(p[3].split()[2])))
for (lab, rays), v in sorted(d.items(), key=lambda kv: (kv[0][1], kv[0][0])):
  print(f'# mean {lab:40s} rays {rays:5d}: {sum(a for a, _ in v) / len(v):.3f} ms, full objective {sum(b for _, b in v) / len(v):.3f} ms')
PY
