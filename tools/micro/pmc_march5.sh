#!/bin/bash
# Counter passes over tools/micro/march5_exp (sel = 1: TS=4 base, TS=7 base, TS=7 hd2+rcp-unchecked), one --pmc group per run.
# Usage (on the GPU box, from the repo root): bash tools/micro/pmc_march5.sh <outdir>
set -u
out=${1:-gpurun_out/r03/pmc_march5}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL" \
           "TCP_PENDING_STALL_CYCLES TCC_TAG_STALL" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i + 1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- tools/micro/march5_exp 4096 5 0 1 > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "march5" not in n: continue
        key = n[n.index("march5"):n.index("march5") + 22]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
ctrs = sorted({c for k in acc for c in acc[k]})
with open(out + "/summary.txt", "w") as fh:
    fh.write("counter".ljust(32) + "".join(n.ljust(26) for n in names) + "\n")
    for c in ctrs:
        fh.write(c.ljust(32) + "".join((("%.4g" % (sum(acc[n][c]) / len(acc[n][c]))) if acc[n][c] else "-").ljust(26) for n in names) + "\n")
print(open(out + "/summary.txt").read())
PY
