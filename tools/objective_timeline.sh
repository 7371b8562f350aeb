#!/bin/bash
# GPU box: what every kernel of the 4096-ray step under the FULL nerf_ds.gin objective costs ALONE (NERFDS_TRAIN_SIDE_STREAMS=0: every launch on one
# stream), summed per kernel name over the last step.  usage: tools/objective_timeline.sh <tag> [case substring]  -> gpurun_out/objective_timeline_<tag>.txt
TAG=${1:-tl}; CASE=${2:-second-order}
OUT=$GRAFT_REPO_ROOT/gpurun_out/otl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NERFDS_TRAIN_SIDE_STREAMS=0 timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/objective_one.py "$CASE" 4 > $OUT/trace.log 2>&1
python - $OUT <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/objective_timeline_$TAG.txt
import csv, glob, sys, os, collections
f = glob.glob(os.path.join(sys.argv[1], 'trace', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_coarse_z' in r['Kernel_Name']]
last = rows[starts[-1]:]
t0, t1 = int(last[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in last)
agg = collections.OrderedDict()
for r in last:
  k = r['Kernel_Name'][:110]
  a = agg.setdefault(k, [0, 0.0])
  a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f'# one step, every launch on ONE stream: span {(t1 - t0) / 1e3:.0f} us, sum of kernel durations {tot:.0f} us, {len(last)} launches')
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
  print(f'{us:9.1f} us  {100 * us / tot:5.1f} %  calls={n:3d}  {k}')
PY
rm -rf $OUT/trace; head -45 $GRAFT_REPO_ROOT/gpurun_out/objective_timeline_$TAG.txt | cut -c1-200
