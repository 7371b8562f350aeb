"""Interleaved A/B timing of library variants in ONE process group of runs (development aid)."""
import os, subprocess, sys
libs = sys.argv[1:]
rounds = 3
res = {l: [] for l in libs}
for r in range(rounds):
  for l in libs:
    env = dict(os.environ, NERFDS_LIB=os.path.abspath(l))
    out = subprocess.run([sys.executable, 'tools/quick_time.py', '65536', 'bf16'], env=env, capture_output=True, text=True, timeout=90).stdout
    ms = [float(x.split()[2]) for x in out.strip().splitlines() if 'R=' in x]
    res[l].append(min(ms[1:]) if len(ms) > 1 else float('nan'))
for l in libs:
  print(os.path.basename(l), ' '.join(f'{x:.2f}' for x in res[l]), 'ms; min', min(res[l]))
