"""Stand-in for the third-party `torch_scatter` package (absent from this image).

TEST INFRASTRUCTURE ONLY. Used by oracle/gen_golden.py and the `reference-live`
tests so that the *unmodified* reference modules under /root/reference/pytorch
can be imported; never imported by the product package.

Semantics restated from the torch_scatter documentation (the reference pins no
version: /root/reference/pytorch/requirements.txt:2) exactly as the reference's
call sites use them: 1-D `index` broadcast along `dim`,
`dim_size` rows in the output, reduce in {sum, mean, max, min}; empty segments
yield 0 for every reduce.  `sum` accumulates sequentially in edge order (CPU
`scatter_add_`), which is the order the HIP kernels reproduce.
"""
import torch


def _expand_index(index, src, dim):
    if index.dim() == src.dim():
        return index
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if dim < 0:
        dim = src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    idx = _expand_index(index, src, dim)
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    if reduce in ("sum", "add"):
        return res.scatter_add_(dim, idx, src)
    if reduce == "mean":
        res.scatter_add_(dim, idx, src)
        cnt = torch.zeros(shape, dtype=src.dtype, device=src.device)
        cnt.scatter_add_(dim, idx, torch.ones_like(src))
        return res / cnt.clamp(min=1)
    if reduce in ("max", "min"):
        return res.scatter_reduce_(dim, idx, src, "amax" if reduce == "max" else "amin",
                                   include_self=False)
    raise ValueError(reduce)
