"""Host-side helpers shared by the nn.Module mirror of the reference's operator surface."""
import torch

from . import _lib
from .plan import plan_for


def scalar0(g):
    """float(g.reshape(-1)[0]) -- the relaxation weight w = g[0] the kernels take by value (JacobiGNN.py:108).  For a GPU
    tensor that read is a device-to-host copy which drains the stream on every call; the value is remembered ON the tensor
    object (attribute `_gla_scalar0`, keyed on its storage pointer, offset, dtype and in-place version counter), so a solver
    loop that passes the same unmodified `g` reads it once.  The cache lives and dies with the tensor: a fresh tensor -- even
    one the allocator places at a recycled address, with version 0 again -- never sees another tensor's value."""
    if not g.is_cuda:
        return float(g.reshape(-1)[0])
    key = (g.untyped_storage().data_ptr(), g.storage_offset(), g.dtype, g._version)
    hit = getattr(g, "_gla_scalar0", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = float(g.reshape(-1)[0])
    try:
        g._gla_scalar0 = (key, val)
    except Exception:                                # (a tensor subclass that refuses attributes: read every time)
        pass
    return val


def compute_device(*tensors):
    """GPU the call runs on: the device of the first GPU operand, else the current HIP device.
    The reference is CPU-only, so CPU operands are accepted, moved once, and results moved back."""
    _lib.require_gpu()
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(t, dev):
    """Move to `dev`, caching the copy on the source tensor so that plans cached on the copy survive
    repeated calls with the same CPU tensor."""
    if t is None or t.device == dev:
        return t
    key = (t._version, dev)
    cached = getattr(t, "_gla_dev_copy", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    c = t.to(dev)
    try:
        t._gla_dev_copy = (key, c)
    except Exception:
        pass
    return c


def back(t, like):
    """Return `t` on the device of the caller's operand `like`."""
    if t is None or not isinstance(like, torch.Tensor) or t.device == like.device:
        return t
    return t.to(like.device)


def float_dtype(*tensors):
    dt = tensors[0].dtype
    if dt not in (torch.float32, torch.float64):
        raise TypeError(f"vertex/edge attributes must be float32 or float64, got {dt}")
    return dt


def cols(t, idx):
    """Column(s) `idx` of an AoS attribute tensor as a contiguous [n, K] operand.
    t is [n, a] (K = 1) or [n, a, K] (multi-RHS extension: trailing dimension K)."""
    if t.dim() == 2:
        return t[:, idx].reshape(-1, 1).contiguous()
    if t.dim() == 3:
        return t[:, idx, :].contiguous()
    raise ValueError(f"attribute tensor must be [n, a] or [n, a, K], got {tuple(t.shape)}")


def split_cols(t):
    """All attribute columns of a 2-D AoS tensor [n, a] as contiguous [n, 1] operands in ONE device pass
    (gla_split_columns); 3-D multi-RHS tensors [n, a, K] fall back to per-column copies."""
    if t.dim() != 2 or not t.is_cuda or t.shape[1] > 8 or t.stride(1) != 1 or t.dtype not in (torch.float32, torch.float64):
        return [cols(t, k) for k in range(t.shape[1])]
    import ctypes
    n, a = t.shape
    outs = [torch.empty((n, 1), dtype=t.dtype, device=t.device) for _ in range(a)]
    arr = (ctypes.c_void_p * a)(*[o.data_ptr() for o in outs])
    from .plan import _DeviceGuard
    with _DeviceGuard(t.device):
        _lib.check(_lib.lib().gla_split_columns(_lib.dtype_code(t.dtype), n, a, _lib.ptr(t),
                                                t.stride(0) if n > 1 else a, arr, _lib.stream_ptr(t.device)))
    return outs


def stack_cols(parts, like):
    """Inverse of `cols`: list of [n, K] -> [n, a] (like.dim()==2) or [n, a, K]."""
    if like.dim() == 2:
        return torch.cat(parts, 1)
    return torch.stack(parts, 1)


def edge_cols(edge_attr, c):
    """edge_attr column 0 (A_ij) next to freshly computed per-edge products c [nnz, K] -> [nnz, 1+K]."""
    A = edge_attr[:, 0].reshape(-1, 1) if edge_attr.dim() == 2 else edge_attr.reshape(-1, 1)
    return torch.cat([A, c], 1)


def aggregate(edgeij_pair, src, n_vertices, reduce="sum"):
    """The reference's aggregation primitive, torch_scatter.scatter(src, edgeij_pair[0], dim=0, dim_size=n,
    reduce=...), on the cached CSR plan of `edgeij_pair` (libgla_hip gla_segment_reduce: rows visited in edge order,
    so sums equal the sequential CPU scatter-add; rows without edges give 0).  src: [nnz] or [nnz, k]."""
    dev = compute_device(src, edgeij_pair)
    plan = plan_for(to_dev(edgeij_pair, dev), n_vertices)
    return back(plan.segment_reduce(to_dev(src, dev), reduce), src)


__all__ = ["aggregate", "split_cols", "compute_device", "to_dev", "back", "float_dtype", "cols", "stack_cols", "edge_cols", "plan_for"]
