#!/usr/bin/env python3
"""fp64 V-cycle fixtures from the UNMODIFIED reference (TEST INFRASTRUCTURE; needs /root/reference).

pytorch/VCycle.py is an fp32 script: it hard-codes `dtype=torch.float` (VCycle.py:83,101,117,126,165,259) and a
module-level N = 5 (:239).  To drive ITS OWN functions (runVCycle / runJacobi / runCheby / runSOC / runDirectInterp /
runResidual, VCycle.py:58-237) in double precision without touching the file, the script is executed once through
oracle/ref_harness.py and its functions are then called with

  * the name `torch.float` rebound to torch.float64 and torch's default dtype set to float64 for the duration of the
    calls (every `dtype=torch.float`, `torch.eye`, `torch.tensor(0.7)` of the script then yields fp64), and
  * the script's global N set to the grid size of the case (runJacobi / runVCycle read it, VCycle.py:165,206),

for N in {5, 8, 16}, 5 cycles V(3,3), 5 coarse Chebyshev steps requested (the script's ChebyGNN has degree 4), the
deterministic stand-in splitting s[0::2] = 1.  Also adds what round 1's fixture lacked: MatVecGNN
results at BASELINE config 1's size (N = 64, k in {1, 2, 16}).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vcycle64.py     ->  tests/golden/vcycle64_vectors.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 24601


def npy(t):
    return t.detach().cpu().numpy()


def main():
    out = {}
    g = ref_harness.run_vcycle_script(SEED, torch)          # fp32 script run (defines the functions + GNN objects)
    ref = ref_harness.load_reference()
    saved_float, saved_default = torch.float, torch.get_default_dtype()
    try:
        torch.float = torch.float64
        torch.set_default_dtype(torch.float64)
        for N in (5, 8, 16):
            g["runVCycle"].__globals__["N"] = N          # the dict the script's functions really read (run_path returns a copy)
            n = N * N
            ei, ev = g["laplacianfun_torch"](N)
            A = torch.sparse_coo_tensor(ei, ev.flatten(), dtype=torch.float64)
            gen = torch.Generator().manual_seed(SEED + 100 + N)
            x = torch.rand((n, 1), generator=gen, dtype=torch.float64)
            b = torch.rand((n, 1), generator=gen, dtype=torch.float64)
            out[f"N{N}_x0"], out[f"N{N}_b"] = npy(x), npy(b)
            out[f"N{N}_res0"] = np.array(float(torch.norm(g["runResidual"](A, b, x))))
            # the pieces of the first cycle, for adapter-level checks
            S = g["runSOC"](A)
            P = g["runDirectInterp"](A, S, N)
            out[f"N{N}_S"] = npy(S).astype(np.uint8)
            out[f"N{N}_P"] = npy(P.to_dense())
            out[f"N{N}_Ac"] = npy((P.t() @ (A @ P)).to_dense())
            out[f"N{N}_jacobi3"] = npy(g["runJacobi"](3, 0.7, A, b, x))
            out[f"N{N}_cheby"] = npy(g["runCheby"](A, b, x, -3.4, -4.0))
            xs, hist = [], []
            for _ in range(5):
                x = g["runVCycle"](A, b, x, 3, 3, 5, True)
                assert x.dtype == torch.float64
                xs.append(npy(x))
                hist.append(float(torch.norm(g["runResidual"](A, b, x))))
            out[f"N{N}_xs"] = np.stack(xs)
            out[f"N{N}_resnorm"] = np.array(hist)
            print(f"N={N}: residual {float(out[f'N{N}_res0']):.3e} -> " + " ".join(f"{h:.3e}" for h in hist))
    finally:
        torch.float = saved_float
        torch.set_default_dtype(saved_default)

    # ---- BASELINE config 1 at its stated size: MatVecGNN on the 64 x 64 Laplacian, k = 1, 2, 16 (MatVecGNN.py:146-150)
    N = 64
    ei, ev = ref.UtilsGNN.laplacianfun_torch(N)
    gen = torch.Generator().manual_seed(SEED + 64)
    X16 = torch.rand(N * N, 16, generator=gen, dtype=torch.float64)
    out["mv64_X16"] = npy(X16)
    mv = ref.MetaLayer(ref.MatVecGNN.EdgeUpdate(), ref.MatVecGNN.VertexUpdate(ref.MatVecGNN.edge_to_vertex_aggregation))
    for k in (1, 2, 16):
        va, ea, _ = mv(X16[:, :k].contiguous(), ei, ev, None, batch=torch.zeros(N * N))
        out[f"mv64_k{k}_va"] = npy(va)
        if k <= 2:
            out[f"mv64_k{k}_ea"] = npy(ea)
    np.savez_compressed(os.path.join(OUT, "vcycle64_vectors.npz"), **out)
    print(f"wrote {len(out)} arrays to {OUT}/vcycle64_vectors.npz "
          f"({os.path.getsize(os.path.join(OUT, 'vcycle64_vectors.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
