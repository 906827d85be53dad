#!/usr/bin/env python3
"""Interleaved A/B timing of the K=1 Jacobi sweep kernel variants (cdna guide rule 24: N variants x M rounds in
ONE process; report median and min).  Usage: python tools/ab_spmv.py [N] [variants...] [--tiles a,b,c]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd._lib import lib, check   # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if args else 4096
    variants = [int(v) for v in args[1:]] or [0, 1, 2, 3, 4]
    tiles = [0]
    for a in sys.argv[1:]:
        if a.startswith("--tiles="):
            tiles = [int(v) for v in a.split("=")[1].split(",")]
    dev = torch.device("cuda:0")
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
    n, nnz = N * N, ei.shape[1]
    B = 12 * nnz + 4 * (n + 1) + 32 * n
    gen = torch.Generator().manual_seed(1)
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
    plans = {}
    for tnz in tiles:
        check(lib().gla_set_option(b"tile_nnz", tnz))
        p = gla.CsrPlan(ei, n)
        p.set_values(ev)
        plans[tnz] = p
    if "--compress" in sys.argv:                       # packed-stream plan (3 B/nnz), timed as tile key -1
        ptiles = [0]
        for a in sys.argv[1:]:
            if a.startswith("--ptiles="):
                ptiles = [int(v) for v in a.split("=")[1].split(",")]
        check(lib().gla_set_option(b"compress", 1))
        for i, pt in enumerate(ptiles):
            check(lib().gla_set_option(b"tile_nnz", pt))
            pc = gla.CsrPlan(ei, n)
            pc.set_values(ev)
            print("packed plan:", bool(pc.info.packed), pc.info.tile_nnz)
            plans[-1 - i] = pc
            tiles = tiles + [-1 - i]
    sweeps, rounds = 10, 7
    ref = None
    res = {}
    for rnd in range(rounds + 1):
        for tnz in tiles:
            for v in variants:
                check(lib().gla_set_option(b"spmv_variant", 0 if v >= 1000 else v))
                check(lib().gla_set_option(b"xcd_chunk", v - 1000 if 1000 <= v < 2000 else 64))   # 1000+C: XCD chunk C
                check(lib().gla_set_option(b"col16", 0 if v == 2000 else 1))     # 2000: CSR kernel, int32 columns
                check(lib().gla_set_option(b"dia", 0 if v in (2000, 3000) else 1))   # 3000: compact CSR kernel, no DIA
                check(lib().gla_set_option(b"compress", 1 if tnz < 0 else 0))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = plans[tnz].jacobi_sweeps(diag, b, x, 0.7, sweeps)
                e1.record()
                torch.cuda.synchronize()
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    elif not torch.equal(out, ref):
                        print(f"MISMATCH variant {v} tile {tnz}: {(out - ref).abs().max().item():.3e}")
                    continue
                res.setdefault((tnz, v), []).append(e0.elapsed_time(e1) * 1e3 / sweeps)
    # box calibration: plain device copy of the same byte volume (read B/2 + write B/2)
    src = torch.empty(B // 16, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)
    cts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        cts.append(e0.elapsed_time(e1) * 1e3)
    cbw = 2 * src.numel() * 8 / sorted(cts)[len(cts) // 2] / 1e3
    print(f"N={N} n={n} nnz={nnz} B={B}  device copy (torch) {cbw:.0f} GB/s")
    for (tnz, v), ts in sorted(res.items()):
        ts = sorted(ts)
        med, mn = ts[len(ts) // 2], ts[0]
        info = plans[tnz].info
        print(f"tile_nnz={info.tile_nnz:5d} lds={info.lds_elems:5d} {'PACKED ' if tnz < 0 else ''}variant={v}: median {med:7.1f} us  min {mn:7.1f} us  "
              f"-> {B / med / 1e3:7.1f} GB/s ({B / med / 8e6 * 100:5.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
