#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_17; mkdir -p $O
export NERFDS_TRAIN_SIDE_STREAMS=0
python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])" >> $O/bench.log 2>&1
bash tools/train_timeline.sh r4serial > /dev/null 2>&1
cat $O/bench.log
python - <<'PY'
import re,collections
rows=[l.split(None,4) for l in open('gpurun_out/timeline_r4serial.txt').read().splitlines()[1:]]
agg=collections.OrderedDict()
for s,e,d,q,k in rows:
    k=re.sub(r'\(.*','',k)[:70]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(d)
tot=sum(v[1] for v in agg.values())
print('total kernel us',round(tot), 'span', rows[-1][1])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]: print(f'{v[1]:9.1f} {v[0]:3d} {k}')
PY
