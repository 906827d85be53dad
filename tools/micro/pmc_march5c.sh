#!/bin/bash
# Counter passes over tools/micro/march5c_exp (sel = 1), one --pmc group per run.
# Usage (on the GPU box, from the repo root): bash tools/micro/pmc_march5c.sh <outdir>
set -u
out=${1:-gpurun_out/r04/pmc_march5c}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i + 1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- tools/micro/march5c_exp 4096 3 0 1 > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "march5c" not in n: continue
        key = n[n.index("march5c"):].split("(")[0][:40]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
ctrs = sorted({c for k in acc for c in acc[k]})
with open(out + "/summary.txt", "w") as fh:
    fh.write("counter".ljust(32) + "".join(n.ljust(42) for n in names) + "\n")
    for c in ctrs:
        fh.write(c.ljust(32) + "".join((f"{sum(acc[n][c]) / max(len(acc[n][c]), 1):.4g}" if acc[n][c] else "-").ljust(42) for n in names) + "\n")
print(open(out + "/summary.txt").read())
PY
