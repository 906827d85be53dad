#!/usr/bin/env python3
"""Where does the construction of dist.RowBlockJacobi spend its time?  W gloo ranks sharing one GPU, an N x (W N) Laplacian
split in W row blocks; rank 0 runs the constructor under cProfile and prints the top entries.
Usage: [GLA_NX=8192] python tools/time_dist_setup.py [N=4096 lines per rank] [W=2]"""
import cProfile, os, pstats, sys, time
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, W, N, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
    dev = torch.device("cuda:0")
    Nx = int(os.environ.get("GLA_NX", N))                   # (GLA_NX: grid-line length, N lines per rank)
    n_local = Nx * N
    lo, hi = rank * n_local, (rank + 1) * n_local
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=N * W, row_range=(lo, hi), device=dev)
    ev = torch.cat([ev, torch.zeros_like(ev)], 1)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.time()
    pr = cProfile.Profile() if rank == 0 else None
    if pr: pr.enable()
    drv = RowBlockJacobi(ei, ev, Nx * N * W, lo, hi)
    torch.cuda.synchronize()
    if pr: pr.disable()
    dist.barrier()
    if rank == 0:
        print(f"constructor: {time.time() - t0:.2f} s; route: {drv.mode}")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    dist.destroy_process_group()


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mp.spawn(worker, args=(W, N, 29533), nprocs=W, join=True)
