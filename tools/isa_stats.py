#!/usr/bin/env python3
"""Static ISA statistics of the kernels in a hipcc .s file (hipcc ... -save-temps=obj): per kernel the register /
spill / scratch metadata and the instruction mix normalised per MFMA.  The numbers committed under profiles/ come
from this script.

  usage: isa_stats.py file.s [substring-of-kernel-name ...] [--top N] [--json out.json]
"""
import collections
import json
import re
import sys

GROUPS = [
    ('mfma', re.compile(r'^v_mfma_')),
    ('valu', re.compile(r'^v_(?!mfma_)')),
    ('salu', re.compile(r'^s_(?!waitcnt|nop|barrier|load|buffer_load|sleep|setprio|endpgm|branch|cbranch)')),
    ('smem', re.compile(r'^s_(load|buffer_load)')),
    ('s_waitcnt', re.compile(r'^s_waitcnt')),
    ('s_nop', re.compile(r'^s_nop')),
    ('s_barrier', re.compile(r'^s_barrier')),
    ('branch', re.compile(r'^s_(branch|cbranch)')),
    ('lds', re.compile(r'^ds_')),
    ('lds_dma', re.compile(r'^buffer_load_.* lds$|^global_load_lds')),
    ('vmem', re.compile(r'^(buffer_|global_|flat_)')),
    ('scratch', re.compile(r'^scratch_')),
]


def kernels(path):
  """yield (name, [instruction lines], metadata dict)"""
  text = open(path).read().splitlines()
  meta = {}
  cur = None
  for l in text:
    m = re.match(r'\s+\.name:\s+(\S+)', l)
    if m and not m.group(1).startswith('ka'):
      cur = m.group(1)
      meta[cur] = {}
      continue
    m = re.match(r'\s+\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|agpr_count|private_segment_fixed_size|'
                 r'group_segment_fixed_size|wavefront_size|max_flat_workgroup_size):\s+(\d+)', l)
    if m and cur:
      meta[cur][m.group(1)] = int(m.group(2))
  body = collections.OrderedDict()
  name = None
  for l in text:
    m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', l)
    if m and m.group(1) in meta:
      name = m.group(1)
      body[name] = []
      continue
    if name is None:
      continue
    if l.startswith('.Lfunc_end') or l.strip().startswith('.end_amdhsa_kernel'):
      name = None
      continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
      continue
    body[name].append(s.split(';')[0].strip())
  for k, v in body.items():
    yield k, v, meta[k]


def classify(op_line):
  for g, rx in GROUPS:
    if rx.match(op_line):
      return g
  return 'other'


def stats(ins, top):
  ops = collections.Counter(i.split()[0] for i in ins)
  grp = collections.Counter()
  for i in ins:
    g = classify(i)
    # LDS-DMA shows up as buffer_load ... lds: classify on the whole line
    grp[g] += 1
  n_mfma = max(1, grp['mfma'])
  out = {'instructions': len(ins), 'mfma': grp['mfma'],
         'per_mfma': {g: round(c / n_mfma, 3) for g, c in sorted(grp.items()) if g != 'mfma'},
         'non_mfma_issue_per_mfma': round((len(ins) - grp['mfma']) / n_mfma, 3),
         'top_ops_per_mfma': {o: round(c / n_mfma, 3) for o, c in ops.most_common(top)}}
  for o in ('v_writelane_b32', 'v_readlane_b32', 'scratch_load_dword', 'scratch_store_dword', 'scratch_load_dwordx4',
            'scratch_store_dwordx4', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 'v_accvgpr_mov_b32'):
    if ops.get(o):
      out.setdefault('spill_traffic', {})[o] = ops[o]
  return out


def main(argv):
  top = 24
  js = None
  pats = []
  path = None
  i = 0
  while i < len(argv):
    a = argv[i]
    if a == '--top':
      top = int(argv[i + 1]); i += 2; continue
    if a == '--json':
      js = argv[i + 1]; i += 2; continue
    if path is None:
      path = a
    else:
      pats.append(a)
    i += 1
  res = {}
  for name, ins, meta in kernels(path):
    if pats and not any(p in name for p in pats):
      continue
    res[name] = {'metadata': meta, **stats(ins, top)}
  txt = json.dumps(res, indent=1)
  if js:
    open(js, 'w').write(txt + '\n')
  print(txt)


if __name__ == '__main__':
  main(sys.argv[1:])
