#!/bin/bash
# Development / build check: device asm of every fused kernel the library ships (15 render kernels + 4 training kernel builds), compiled
# with the Makefile's own commands (tools/isa_snapshot.sh), through tools/isa_lint.py.  Exit status 1 when a suspect copy is found.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out=${1:-$ROOT/nerf-ds_amd/csrc/build/asm}
KEEP_RAW=1 "$ROOT/tools/isa_snapshot.sh" "$out" > /dev/null 2>&1 || exit 2
python3 "$ROOT/tools/isa_lint.py" "$out"/raw_*.s | grep -v "^$"
exit ${PIPESTATUS[0]}
