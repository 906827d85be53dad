#!/bin/bash
# SQ / TCP counters of the headline step's three march passes (the module step of tools/pmc_workload.py), one --pmc group per run.
# Usage (GPU box, repo root): bash tools/pmc_headline.sh <outdir>
set -u
out=${1:-gpurun_out/r04/pmc_headline}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace -d "$out/g$i" -o g$i --output-format csv -- python3 tools/pmc_workload.py 4096 > "$out/g$i.log" 2>&1 || echo "group $i ($grp) failed" >> "$out/failed.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(n):
    m = re.search(r"jacobi_march5_kernel<\w+,\s*(\d+),\s*(\d+)[^>]*?,\s*(\d+)>\(", n)
    return f"TS={m.group(1)} MODE={m.group(2)} DGM={m.group(3)}" if m else None
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/g1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
names = sorted(acc)
ctrs = sorted({c for k in acc for c in acc[k]})
with open(out + "/summary.txt", "w") as fh:
    fh.write("per launch (mean over the launches of the counter passes); kernels of the module step, 4096^2, fp64\n")
    fh.write("counter".ljust(32) + "".join(n.ljust(26) for n in names) + "\n")
    for c in ctrs:
        fh.write(c.ljust(32) + "".join((f"{sum(acc[n][c]) / max(len(acc[n][c]), 1):.4g}" if acc[n][c] else "-").ljust(26) for n in names) + "\n")
    fh.write("kernel_us (group 1 run)".ljust(32) + "".join((f"{sorted(dur[n])[len(dur[n]) // 2]:.1f}" if dur[n] else "-").ljust(26) for n in names) + "\n")
    for n in names:
        a = acc[n]
        g = lambda c: sum(a[c]) / len(a[c]) if a[c] else float("nan")
        cyc = g("GRBM_GUI_ACTIVE") / 8                       # summed over the 8 XCDs
        fh.write(f"{n}: {cyc:.0f} cycles per XCD; VALU instructions {g('SQ_INSTS_VALU'):.3g} x 4 cycles / (1024 SIMDs x cycles) = "
                 f"{g('SQ_INSTS_VALU') * 4 / (1024 * cyc):.2f} of the VALU issue slots; waves resident {g('SQ_WAVE_CYCLES') * 4 / (cyc * 2048):.2f} of 2 per SIMD; "
                 f"waiting at s_waitcnt {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f} of the wave time\n")
print(open(out + "/summary.txt").read())
PY
