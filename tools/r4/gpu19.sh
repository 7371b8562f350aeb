#!/bin/bash
# what does the L2 overflow of the split-bf16 weight set cost?  `one` = measurement build whose fine level walks the coarse level's stream
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4_19; mkdir -p $O
L=$PWD/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_one.so
python tools/ab.py bf16x3 4 main $L > $O/ab.txt 2>&1
python tools/ab.py bf16 3 main $L >> $O/ab.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for v in main one; do
  if [ $v = one ]; then export NERFDS_LIB=$L; else unset NERFDS_LIB; fi
  rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $O/pmc_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py 65536 bf16x3 > $O/pmc_$v.log 2>&1
  python - $O/pmc_$v $v <<'PY' >> $O/ab.txt
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f[0])):
    if 'render_rays' in r['Kernel_Name']:
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,v in agg.items(): print(sys.argv[2], k, 'per launch %.4g' % (v[1]/v[0]), 'launches', v[0])
PY
  rm -rf $O/pmc_$v
done
cat $O/ab.txt
