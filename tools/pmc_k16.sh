#!/bin/bash
# Counter passes (one --pmc group per run) over tools/pmc_workload_k16.py: the K = 16 Chebyshev / power / Jacobi march kernels.
# Usage (on the GPU box, from the repo root): bash tools/pmc_k16.sh <outdir>
set -u
out=${1:-gpurun_out/r03/pmc_k16}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- python3 tools/pmc_workload_k16.py > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(power_march_kernel|cheby_march5_kernel|jacobi_marchk_kernel|spmm_dia2d_kernel)<[^(]*", n)
        if not m: continue
        key = m.group(0)[:60]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
ctrs = sorted({c for k in acc for c in acc[k]})
with open(out + "/summary.txt", "w") as fh:
    for n in names: fh.write("# " + n + "\n")
    fh.write("counter".ljust(28) + "".join(("k%d" % i).ljust(16) for i in range(len(names))) + "\n")
    for c in ctrs:
        fh.write(c.ljust(28) + "".join((("%.4g" % (sum(acc[n][c]) / len(acc[n][c]))) if acc[n][c] else "-").ljust(16) for n in names) + "\n")
print(open(out + "/summary.txt").read())
PY
