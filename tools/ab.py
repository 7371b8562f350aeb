"""Interleaved A/B timing of library variants (development aid).  usage: python tools/ab.py <prec> <rounds> lib1.so lib2.so ...
('main' = the in-tree library).  Each (round, lib) is one process running tools/quick_time.py at 65 536 rays; min over the
timed launches of a process, then min / median over rounds."""
import os, subprocess, sys, statistics
prec, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in libs}
for r in range(rounds):
  for l in libs:
    env = dict(os.environ)
    if l != 'main':
      env['NERFDS_LIB'] = os.path.abspath(l)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools/quick_time.py'), '65536', prec], env=env, capture_output=True, text=True, timeout=300)
    ms = [float(x.split()[2]) for x in out.stdout.strip().splitlines() if 'R=' in x]
    if len(ms) < 2:
      print(l, 'FAILED', out.stderr[-400:])
    res[l].append(min(ms[1:]) if len(ms) > 1 else float('nan'))
for l in libs:
  v = res[l]
  print(f'{prec:7s} {os.path.basename(l):28s}', ' '.join(f'{x:.2f}' for x in v), f' ms; min {min(v):.2f} median {statistics.median(v):.2f}  -> {65536 / min(v) / 1e3:.3f} Mrays/s', flush=True)
