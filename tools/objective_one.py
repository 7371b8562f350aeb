"""GPU: N steps of the 4096-ray training step under ONE objective of tools/objective_time.py (for profilers): python tools/objective_one.py <case substring> [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
which, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
src = open(os.path.join(ROOT, 'tools', 'objective_time.py')).read()
head, loop = src.split('tr = Trainer(cfg, params, max_rays=R)')
exec(compile(head, 'objective_time_head', 'exec'))
ob = [v for k, v in cases.items() if which in k][0]
tr = Trainer(cfg, params, max_rays=R)
for _ in range(steps):
  tr.step(batch, EX, 1e-3, objective=ob)
torch.cuda.synchronize()
