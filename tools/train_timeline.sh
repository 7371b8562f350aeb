#!/bin/bash
# GPU box: kernel timeline of ONE training step (start / end of every dispatch, in order), to see what is serial and what overlaps.
# usage: tools/train_timeline.sh <tag>   -> gpurun_out/timeline_<tag>.txt
TAG=${1:-tl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --train --steps 3 --warmup 2 --no-cpu-baseline --no-full-objective --no-option-legs ${TRAIN_ARGS:-} > $OUT/trace.log 2>&1
python - $OUT <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG.txt
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], 'trace', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last k_coarse_z dispatch on
starts = [i for i, r in enumerate(rows) if 'k_coarse_z' in r['Kernel_Name']]
i0 = starts[-1]
t0 = int(rows[i0]['Start_Timestamp'])
print('columns: start_us end_us dur_us queue kernel')
for r in rows[i0:]:
  s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
  print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:90]}")
PY
rm -rf $OUT/trace
