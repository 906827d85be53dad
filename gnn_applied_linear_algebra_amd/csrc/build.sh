#!/bin/bash
# Build libgla_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : no FMA contraction, so row sums and vertex updates round exactly like the
#                       reference's separate torch mul / add ops (bit-exact parity)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result"
OUT=../libgla_hip.so
pids=()
for f in gla_rows gla_edges gla_plan gla_sparse gla_mlp gla_dist gla_cheby gla_jacobik; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ gla_internal.h -nt $f.o ] || [ gla_device.h -nt $f.o ] || [ ../../include/gla.h -nt $f.o ]; then
    rm -f $f.o                                   # a failed compile must not leave a stale object behind
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done          # (set -e: a failed translation unit stops the build)
$HIPCC --offload-arch=gfx950 -shared -fPIC gla_rows.o gla_edges.o gla_plan.o gla_sparse.o gla_mlp.o gla_dist.o gla_cheby.o gla_jacobik.o -ldl -o $OUT
echo "built $(realpath $OUT)"
