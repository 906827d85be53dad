#!/bin/bash
# Scan the gfx950 code objects of libgla_hip.so's translation units (the .o files csrc/build.sh leaves) for the signature of
# the hipcc 7.0 miscompile recorded in profiles/r03/marchk_miscompile.txt: an exec-masked region with nothing in it
# (s_and_saveexec immediately followed by the restoring s_or_b64 exec) -- the masked move of a `uniform || per_lane ? value : 0`
# select dropped.  Prints a count per file; exits non-zero if any is found.   Usage: tools/check_isa.sh [file.o ...]
set -uo pipefail
src="$(cd "$(dirname "$0")/../gnn_applied_linear_algebra_amd/csrc" && pwd)"
OBJDUMP=${OBJDUMP:-/opt/rocm/lib/llvm/bin/llvm-objdump}
files=("$@"); [ ${#files[@]} -eq 0 ] && files=("$src"/gla_*.o)
tmp=$(mktemp -d); bad=0
for f in "${files[@]}"; do
  b=$(basename "$f")
  cp "$f" "$tmp/$b"
  ( cd "$tmp" && $OBJDUMP --offloading "$b" > /dev/null 2>&1 )
  co=$(ls "$tmp/$b".*gfx950 2>/dev/null | head -1)
  if [ -z "$co" ]; then echo "$b: no gfx950 code object"; bad=1; continue; fi
  n=$($OBJDUMP -d "$co" 2>/dev/null | grep -A1 's_and_saveexec_b64' | grep -B1 's_or_b64 exec' | grep -c saveexec)
  echo "$b: $n empty exec-masked regions"
  [ "$n" -eq 0 ] || bad=1
done
rm -rf "$tmp"
exit $bad
