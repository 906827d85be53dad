#!/bin/bash
# SQ counters of the K-wide Jacobi pass (tools/pmc_marchk_workload.py), one --pmc group per run.
# Usage (GPU box, repo root): bash tools/pmc_marchk.sh <outdir>
set -u
out=${1:-gpurun_out/r05/pmc_marchk}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i + 1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- python3 tools/pmc_marchk_workload.py 4096 > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jacobi_marchk_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/g1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jacobi_marchk_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/summary.txt", "w") as fh:
    for c in sorted(acc):
        fh.write(f"{c:32s} {sum(acc[c]) / len(acc[c]):.4g}\n")
    if dur:
        fh.write(f"kernel_us (group 1 run, median)  {sorted(dur)[len(dur) // 2]:.1f}\n")
print(open(out + "/summary.txt").read())
PY
