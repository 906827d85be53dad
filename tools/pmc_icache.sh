#!/bin/bash
# Instruction-cache counters of the headline step's march passes (tools/pmc_workload.py), one --pmc group per run.
# Usage (GPU box, repo root): bash tools/pmc_icache.sh <outdir>
set -u
out=${1:-gpurun_out/r05/pmc_icache}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i + 1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- python3 tools/pmc_workload.py 4096 > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def key(n):
    m = re.search(r"jacobi_march5_kernel<\w+,\s*(\d+),\s*(\d+)", n)
    return f"TS={m.group(1)} MODE={m.group(2)}" if m else None
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(acc):
        fh.write(k + ": " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items())) + "\n")
print(open(out + "/summary.txt").read())
PY
