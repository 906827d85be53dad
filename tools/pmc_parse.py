#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs of tools/pmc_workload.py (gpurun_out/pmc/{fetch,write,l2}) to
profiles/traffic_jacobi.json + a text summary.  Units and corrections per MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of a wide streaming read -> x2, checked here
against a calibration kernel of known volume (torch a + c on 2 x 512 MiB)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
col16 = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # the workload ran with int16 tile-relative columns
dia = int(sys.argv[5]) if len(sys.argv) > 5 else 5          # diagonals of the stencil layout (0: CSR kernels)
import re
res = collections.defaultdict(lambda: collections.defaultdict(list))
kernel_names = set()


def sweeps_of(name):
    """sweeps per launch of a Jacobi kernel name (temporal blocking: jacobi_march5_kernel<T, TS, ..>, jacobi_fused_dia2d_kernel<T, D, TS, ..>)"""
    m = re.search(r"jacobi_march5_kernel<\w+,\s*(\d+)", name) or re.search(r"jacobi_fused_dia2d_kernel<\w+,\s*\d+,\s*(\d+)", name)
    return int(m.group(1)) if m else 1


rows = []
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
# the workload's sweeps may be a mix of passes (20 = 7 + 7 + 6): keep the dominant kernel (the one covering most sweeps)
cover = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    if "EpiJacobi" in name or "jacobi_fused_dia2d" in name or "jacobi_march5" in name:
        cover[name] += sweeps_of(name)
dominant = cover.most_common(1)[0][0] if cover else None
for r in rows:
    name = r["Kernel_Name"]
    key = "jacobi" if name == dominant else ("calib_add" if "CUDAFunctor_add" in name else None)
    if key:
        res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if key == "jacobi":
            kernel_names.add(name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:120])
mean = lambda v: sum(v) / len(v)
j, c = res["jacobi"], res["calib_add"]
calib_read = 2 * 512 * 1024                      # KiB actually read by the calibration kernel
fetch_factor = calib_read / mean(c["FETCH_SIZE"])
write_factor = 512 * 1024 / mean(c["WRITE_SIZE"])
n, nnz = N * N, 5 * N * N - 4 * N
alg = 12 * nnz + 4 * (n + 1) + 32 * n
layout = (10 * nnz + 2 * (n + n // 256) + 32 * n) if col16 else alg      # int16 columns + uint16 row offsets
if dia:
    layout = (8 * dia + 2 + 32) * n                                        # diagonal-major values + row mask
sweeps_per_launch = sweeps_of(dominant) if dominant else 1
fused = sweeps_per_launch > 1                                              # several sweeps per launch (temporal blocking)
alg *= sweeps_per_launch
if fused:
    layout = (8 * dia + 2 + 32) * n                                        # ONE pass over the data per launch
rd = mean(j["FETCH_SIZE"]) * 1024 * 2.0
wr = mean(j["WRITE_SIZE"]) * 1024
out = {"n": N, "col16": col16, "dia": dia,
       "kernel": sorted(kernel_names)[0] if kernel_names else None, "sweeps_per_launch": sweeps_per_launch,
       "launches": len(j["FETCH_SIZE"]),
       "FETCH_SIZE_KiB": mean(j["FETCH_SIZE"]), "WRITE_SIZE_KiB": mean(j["WRITE_SIZE"]),
       "fetch_correction": 2.0, "calibration_fetch_factor_measured": round(fetch_factor, 4),
       "calibration_write_factor_measured": round(write_factor, 4),
       "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
       "hbm_bytes_per_launch": int(rd + wr), "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": round((rd + wr) / alg, 4),
       "layout_bytes_per_launch": layout, "traffic_over_layout": round((rd + wr) / layout, 4),
       "TCC_HIT_sum": mean(j["TCC_HIT_sum"]) if j.get("TCC_HIT_sum") else None,
       "TCC_MISS_sum": mean(j["TCC_MISS_sum"]) if j.get("TCC_MISS_sum") else None}
os.makedirs(os.path.join(ROOT, "profiles", tag), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_jacobi.json"), "w"), indent=1)
json.dump(out, open(os.path.join(ROOT, "profiles", tag, "pmc_traffic_jacobi.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
