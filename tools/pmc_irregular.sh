#!/bin/bash
# Where the time of the K = 1 CSR SpMV goes on irregular matrices: HBM bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) against the
# requests the x gather sends from the L1s to the L2 (TCP_TCC_READ_REQ; one request moves a 64-byte... see the summary).
# Usage (GPU box, repo root): bash tools/pmc_irregular.sh <outdir>
set -u
out=${1:-gpurun_out/r04/pmc_irregular}
mkdir -p "$out"
export TMPDIR=/tmp
for kind in laplacian_csr powerlaw_random powerlaw_banded medium_random medium_banded; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i + 1))
    timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace -d "$out/$kind.g$i" -o p --output-format csv -- python3 tools/pmc_workload_irregular.py $kind > "$out/$kind.g$i.log" 2>&1 || echo "$kind group $i ($grp) failed" >> "$out/failed.txt"
  done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in glob.glob(out + "/*.g*"):
    if not os.path.isdir(d): continue
    kind = os.path.basename(d).split(".")[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmv_stream_vec2_kernel" in r["Kernel_Name"] or "spmv_stream_kernel" in r["Kernel_Name"]:
                acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmv_stream_vec2_kernel" in r["Kernel_Name"]:
                dur[kind].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
kinds = ["laplacian_csr", "powerlaw_random", "powerlaw_banded", "medium_random", "medium_banded"]
ctrs = sorted({c for k in acc for c in acc[k]})
with open(out + "/summary.txt", "w") as fh:
    fh.write("per launch (mean over the launches of the counter passes; kernel time under the profiler)\n")
    fh.write("counter".ljust(34) + "".join(k.ljust(20) for k in kinds) + "\n")
    for c in ctrs:
        fh.write(c.ljust(34) + "".join((f"{sum(acc[k][c]) / len(acc[k][c]):.4g}" if acc[k][c] else "-").ljust(20) for k in kinds) + "\n")
    fh.write("kernel_us".ljust(34) + "".join((f"{sorted(dur[k])[len(dur[k]) // 2]:.1f}" if dur[k] else "-").ljust(20) for k in kinds) + "\n")
    for k in kinds:
        a = acc[k]
        if a.get("FETCH_SIZE") and a.get("WRITE_SIZE"):
            hbm = (2 * sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"]) + sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"])) * 1024
            fh.write(f"{k}: HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB units) = {hbm / 1e6:.1f} MB")
            if a.get("TCP_TCC_READ_REQ_sum"):
                rq = sum(a["TCP_TCC_READ_REQ_sum"]) / len(a["TCP_TCC_READ_REQ_sum"])
                fh.write(f"; L1 -> L2 read requests {rq / 1e6:.2f} M")
            fh.write("\n")
print(open(out + "/summary.txt").read())
PY
