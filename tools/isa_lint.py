#!/usr/bin/env python3
"""Development lint for the hipcc issue described at eval_shared (field.h): reports vector copies/spills that
sit at the top of a basic block BEFORE the instruction that restores exec (s_or_b64 exec, exec, ...), i.e. copies that
run under the partial EXEC of the region being closed.   usage: hipcc ... --cuda-device-only -S -o k.s ; isa_lint.py k.s"""
import re, sys

SUSPECT = re.compile(r'\s+(v_accvgpr_write_b32|v_accvgpr_read_b32|v_mov_b32|v_mov_b64|scratch_store|scratch_load|buffer_store_dword .*offen|v_accvgpr_mov)')
RESTORE = re.compile(r'\s+s_or_b64 exec, exec,')   # structured-CF join (s_or_saveexec -1 / s_mov exec pairs are WWM spill code: fine)

def main(path):
  lines = open(path).read().splitlines()
  bad = 0; regions = 0
  i = 0
  while i < len(lines):
    if re.match(r'^\.LBB\d+_\d+:', lines[i]):
      j = i + 1; pre = []
      while j < len(lines) and j < i + 200:   # (SGPR-spill writelanes and scalar loads can put 20+ lines ahead of the restore)
        l = lines[j]
        if l.strip().startswith(';') or not l.strip():
          j += 1; continue
        if RESTORE.match(l):
          regions += 1
          for k, t in pre:
            if SUSPECT.match(t):
              bad += 1
              print(f'{path}:{k + 1}: {t.strip()}   (before exec restore at line {j + 1})')
          break
        if re.match(r'^\S', l) or l.strip().startswith('s_cbranch') or l.strip().startswith('s_branch') or 'saveexec' in l or re.match(r'\s+s_\w+_b64 exec,', l):   # a region opens: what follows is its body
          break
        pre.append((j, l)); j += 1
    i += 1
  print(f'{path}: {regions} exec-restoring join blocks, {bad} suspect copies ahead of the restore')
  return 1 if bad else 0

if __name__ == '__main__':
  sys.exit(max(main(p) for p in sys.argv[1:]))
