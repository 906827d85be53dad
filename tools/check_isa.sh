#!/bin/bash
# Scan the gfx950 ISA of every translation unit of libgla_hip.so for the signature of the hipcc 7.0 miscompile recorded in
# profiles/r03/marchk_miscompile.txt: an exec-masked region with nothing in it (s_and_saveexec immediately followed by the
# restoring s_or_b64 exec) -- the masked move of a `uniform || per_lane ? value : 0` select dropped.  Prints a count per file;
# exits non-zero if any is found.   Usage: tools/check_isa.sh [file.hip ...]
set -uo pipefail
cd "$(dirname "$0")/../gnn_applied_linear_algebra_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -S --cuda-device-only"
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(gla_*.hip)
tmp=$(mktemp -d); bad=0
for f in "${files[@]}"; do
  ( $HIPCC $FLAGS "$f" -o "$tmp/${f%.hip}.s" 2>/dev/null
    n=$(grep -A1 's_and_saveexec_b64' "$tmp/${f%.hip}.s" | grep -B1 's_or_b64 exec' | grep -c saveexec)
    echo "$f: $n empty exec-masked regions"; [ "$n" -eq 0 ] ) &
done
for j in $(jobs -p); do wait "$j" || bad=1; done
rm -rf "$tmp"
exit $bad
