"""profiles/<tag>_hbm_traffic.json from a tools/prof_bench.sh summary: HBM bytes per launch (FETCH_SIZE x 2 per MI355X_MICROARCH.md + WRITE_SIZE,
both in KiB), TCC hit rate, matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCD-copies)).
usage: python tools/traffic_from_summary.py <summary.txt> <kernel label> <source path recorded in the json> [round label] > out.json"""
import json, re, sys
txt = open(sys.argv[1]).read()
def per(name):
  m = re.search(r'pmc ' + name + r'\s+sum=\S+ dispatches=\d+ per_dispatch=(\S+)', txt)
  return float(m.group(1)) if m else None
fetch, write, hit, miss = per('FETCH_SIZE'), per('WRITE_SIZE'), per('TCC_HIT_sum'), per('TCC_MISS_sum')
busy, gui = per('SQ_VALU_MFMA_BUSY_CYCLES'), per('GRBM_GUI_ACTIVE')
out = {'kernel': sys.argv[2], 'source': sys.argv[3], 'fetch_size_kb_per_launch': fetch, 'write_size_kb_per_launch': write,
       'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024, 'tcc_hit_rate': hit / (hit + miss) if hit and miss else None,
       'mfma_busy_frac': busy / (1024 * gui / 8) if busy and gui else None,
       'measured_on': (sys.argv[4] if len(sys.argv) > 4 else 'round 3 (final build)') + ', separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline '
                      '--no-other-paths` (tools/prof_bench.sh), FETCH_SIZE doubled per MI355X_MICROARCH.md, per launch of 60000 rays on average'}
print(json.dumps(out, indent=1))
