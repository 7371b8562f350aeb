#!/bin/bash
# Development: variant libraries that differ from the in-tree build in the compile flags of some objects (A/B timing with tools/ab.py, tools/ab_train.sh).
#   usage: tools/variant.sh <name>:<object>[,<object>...]:"<extra flags>" ...        (needs a finished `make`)
#     object = a csrc/build object without .o: k_nerfds_bf16x3, k_train_fwd16, host, train_k ...
#   result: nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_<name>.so ; run with NERFDS_LIB=<that file>
# The compile and link commands come from `make -n`, so a variant is the shipped build plus the extra flags (later -D wins).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/nerf-ds_amd/csrc" || exit 1
mkdir -p build/abl ../nerfds_amd/_lib/abl
make -n -B all 2>/dev/null > build/abl/cmds.txt
for v in "$@"; do
  n=${v%%:*}; r=${v#*:}; objs=${r%%:*}; f=${r#*:}
  (
    link=$(grep -E -- "-shared .*libnerfds_hip.so" build/abl/cmds.txt | sed -E "s#-o [^ ]*libnerfds_hip.so#-o $ROOT/nerf-ds_amd/nerfds_amd/_lib/abl/libnerfds_hip_$n.so#")
    for o in ${objs//,/ }; do
      cmd=$(grep -E -- "-o [^ ]*/build/$o\.o\$" build/abl/cmds.txt | head -1)
      [ -z "$cmd" ] && { echo "$n: no object $o in the Makefile"; exit 1; }
      cmd=$(echo "$cmd" | sed -E "s# -o [^ ]*/build/$o\.o\$# $f -Rpass-analysis=kernel-resource-usage -o $ROOT/nerf-ds_amd/csrc/build/abl/${o}_$n.o#")
      bash -c "$cmd" 2>&1 | grep -E "error|VGPRs Spill|ScratchSize|Occupancy" | sort | uniq -c | sed "s/^/$n $o: /" &
      link=$(echo "$link" | sed -E "s#[^ ]*/build/$o\.o#$ROOT/nerf-ds_amd/csrc/build/abl/${o}_$n.o#")
    done
    wait
    bash -c "$link" 2>&1 | grep -v "hip-link"
  ) &
done
wait
ls -la ../nerfds_amd/_lib/abl/ | tail -n +2
