#!/bin/bash
# Device ISA of every fused-kernel translation unit, exactly as csrc/Makefile compiles them (the compile commands come from `make -n`):
#   tools/isa_snapshot.sh <outdir>      -> <outdir>/k_<name>.s, one per kernel object, comments, directives and the compilation-unit id stripped
# Two snapshots of the same source are byte-identical; used to prove that a refactor changed no instruction (diff -r a b).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out=$(realpath -m "$1"); mkdir -p "$out"
cd "$ROOT/nerf-ds_amd/csrc"
make -n -B all 2>/dev/null | grep -E "hipcc.* -c [a-z_]*kernel[a-z_]*\.hip .*-DNERFDS_NAME" | while read -r cmd; do
  name=$(echo "$cmd" | sed -E 's/.*-DNERFDS_NAME=([a-z0-9_]+).*/\1/')
  echo "$cmd" | sed -E "s# -c # --cuda-device-only -S #; s# -o [^ ]+# -o $out/raw_$name.s#"
done > "$out/cmds.txt"
xargs -P ${JOBS:-8} -I{} bash -c "{}" < "$out/cmds.txt"
for f in "$out"/raw_*.s; do
  n=$(basename "$f" | sed 's/^raw_//')
  # instructions and labels only: no comments, no assembler directives (metadata carries source paths and compiler ids)
  sed -E 's/;.*$//; s/[ \t]+$//' "$f" | grep -vE '^\s*\.' | grep -vE '^\s*$' | grep -v '^__hip_cuid_' > "$out/k_$n"
  [ -n "$KEEP_RAW" ] || rm "$f"      # KEEP_RAW=1: the raw assembler output too (labels, directives: tools/isa_lint.sh)
done
ls "$out" | wc -l
