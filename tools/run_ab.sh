#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
 for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -q -k deterministic 2>&1 | tail -3; done
 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5
 timeout 900 python bench.py --steps 10 --warmup 2 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bf16', d['value'], d['roofline']['frac'], d.get('rgb_max_rel_err')); print('parity', d['parity_path']['value'], d['parity_path']['rgb_max_rel_err'])
for p in d['other_paths']: print(p['precision'], p['value'], p['rgb_max_rel_err'])"
} > gpurun_out/r2_final_check.log 2>&1
cat gpurun_out/r2_final_check.log
