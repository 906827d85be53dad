import sys, os, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/gnn_applied_linear_algebra_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import gnn_applied_linear_algebra_amd as gla
dev = torch.device("cuda:0")
n, K = 1 << 20, 16
idx = torch.arange(n, device=dev)
plan = gla.CsrPlan(torch.stack([idx, idx]), n)
plan.set_values(torch.ones(n, dtype=torch.float64, device=dev))
gen = torch.Generator().manual_seed(5)
bad = tot = 0
for rnd in range(40):
    X = (torch.rand(n, K, generator=gen, dtype=torch.float64) + 1e-3) * (10.0 ** float(torch.randint(-30, 30, (1,), generator=gen)))
    if rnd % 3 == 0:
        X = -X
    nrm = (torch.rand(K, generator=gen, dtype=torch.float64) + 0.5) * (2.0 ** float(torch.randint(-40, 40, (1,), generator=gen)))
    if rnd == 7:                       # the documented exception: significand all ones
        nrm[0] = float.fromhex("0x1.fffffffffffffp+3")
    Y, _, _ = plan.power_step(X.to(dev), nrm=nrm.to(dev))
    want = X / nrm
    neq = (Y.cpu() != want)
    bad += int(neq.sum()); tot += X.numel()
    if int(neq.sum()):
        cols = neq.any(0).nonzero().flatten().tolist()
        print("round", rnd, "mismatches", int(neq.sum()), "columns", cols, [nrm[c].hex() for c in cols][:3])
print(f"{bad} mismatches in {tot} divisions")
