import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
dev = torch.device("cuda:0")
for N in (4096, 1024):
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
    n = N * N
    gen = torch.Generator().manual_seed(1)
    va = torch.rand(n, 2, generator=gen, dtype=torch.float64).to(dev)
    g = torch.tensor([-3.4, -4.0], dtype=torch.float64)
    for deg in (4, 8):
        cheb = gla.ChebyGNN.ChebyRelaxGNN(deg, return_edge_attr=False)
        res, outs = {}, {}
        for rnd in range(5):
            for name, fuse in (("step by step", 0), ("fused", 1)):
                check(lib().gla_set_option(b"jacobi_fuse", fuse))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): out = cheb(va, ei, ev, g)
                e1.record(); torch.cuda.synchronize()
                if rnd: res.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
                outs[name] = out[0]
        check(lib().gla_set_option(b"jacobi_fuse", 1))
        print(f"ChebyRelaxGNN(deg={deg}) K=1 {N}^2: step by step {sorted(res['step by step'])[2]:.3f} ms, fused {sorted(res['fused'])[2]:.3f} ms, bit-identical {bool(torch.equal(outs['fused'], outs['step by step']))}")
