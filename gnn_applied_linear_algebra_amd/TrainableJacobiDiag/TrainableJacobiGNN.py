"""Learned Jacobi diagonal -- mirror of /root/reference/pytorch/TrainableJacobiDiag/TrainableJacobiGNN.py.

One MetaLayer with a vertex model only (:49-51): the vertex update concatenates the vertex attribute (A_ii) with the
min / mean / sum / max of the row's off-diagonal entries (:53-70, four torch_scatter.scatter calls in the reference,
ONE pass of gla_segment_stats here) and feeds the 5 numbers through a 5-50-20-1 ReLU MLP (:86-91).  With autograd
on (training) the MLP is the plain torch modules (rocBLAS GEMMs); under torch.no_grad() / frozen parameters
(validation, deployment) it is one fused kernel, gla_mlp_relu3, whose hidden layers stay in registers.  The edge
attributes are data, so the aggregation needs no backward.
"""
import torch
from torch.nn import Linear, ReLU, Sequential

from .._common import compute_device, plan_for, to_dev
from ..meta import MetaLayer


def get_model():
    """:49-51"""
    return MetaLayer(None, VertexUpdate(edge_to_vertex_aggregation), None)


def edge_to_vertex_aggregation(edgeij_pair, edge_attr, num_nodes):
    """:53-70  cat(min, mean, sum, max) over the edges leaving each vertex -> [num_nodes, 4 * n_edge_attr];
    vertices without edges get 0 (torch_scatter semantics)."""
    if edge_attr.requires_grad:
        raise ValueError("edge_to_vertex_aggregation: edge attributes are data on this path (no backward kernel)")
    dev = compute_device(edge_attr, edgeij_pair)
    plan = plan_for(to_dev(edgeij_pair, dev), num_nodes)
    out = plan.segment_stats(to_dev(edge_attr, dev))
    return out if edge_attr.is_cuda else out.to(edge_attr.device)


class VertexUpdate(torch.nn.Module):
    """:72-111"""

    def __init__(self, edge_agg):
        super().__init__()
        self.edge_agg = edge_agg
        self.vertex_update = Sequential(Linear(5, 50), ReLU(), Linear(50, 20), ReLU(), Linear(20, 1))
        self.vertex_update.apply(init_weights)

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        agg_edges = self.edge_agg(edgeij_pair, edge_attr, vertex_attr.size(0))
        update_input = torch.cat([vertex_attr, agg_edges.to(vertex_attr.device)], dim=1)
        if not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters()):
            fused = _fused_inference(self.vertex_update, update_input)        # no autograd graph wanted
            if fused is not None:
                return fused
        return self.vertex_update(update_input)


def _fused_inference(seq, x):
    """gla_mlp_relu3 for a Sequential(Linear, ReLU, Linear, ReLU, Linear(.,1)) on GPU operands, else None
    (the torch modules then run as they are -- they are also the training path, with autograd)."""
    mods = list(seq)
    if len(mods) != 5 or not all(isinstance(mods[i], Linear) for i in (0, 2, 4)) or \
            not all(isinstance(mods[i], ReLU) for i in (1, 3)):
        return None
    l1, l2, l3 = mods[0], mods[2], mods[4]
    if not x.is_cuda or x.dim() != 2 or x.dtype not in (torch.float32, torch.float64) or l3.out_features != 1 or \
            l1.in_features > 8 or l1.out_features > 512 or l2.out_features > 32 or \
            any(m.bias is None or m.weight.dtype != x.dtype or m.weight.device != x.device for m in (l1, l2, l3)):
        return None
    from .._lib import check, dtype_code, lib, ptr, stream_ptr
    from ..plan import _DeviceGuard
    x = x.contiguous()
    out = torch.empty((x.shape[0], 1), dtype=x.dtype, device=x.device)
    w = [t.detach().contiguous() for t in (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias)]
    work = torch.empty(lib().gla_mlp_work_bytes(l1.out_features), dtype=torch.uint8, device=x.device)
    with _DeviceGuard(x.device):
        check(lib().gla_mlp_relu3(dtype_code(x.dtype), x.shape[0], l1.in_features, l1.out_features, l2.out_features,
                                  ptr(x), x.stride(0) if x.shape[0] > 1 else l1.in_features, *[ptr(t) for t in w],
                                  ptr(out), ptr(work), stream_ptr(x.device)))
    return out


def init_weights(m):
    """:113-117  weights ~ U[0, 1), biases 0.01"""
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.uniform_(m.weight, a=0.0, b=1.0)
        m.bias.data.fill_(0.01)
