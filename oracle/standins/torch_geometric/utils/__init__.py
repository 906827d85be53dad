"""`torch_geometric.utils` stand-in. TEST INFRASTRUCTURE ONLY."""


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def add_self_loops(*a, **k):  # imported by the reference, never called on the hot path
    raise NotImplementedError


def degree(*a, **k):
    raise NotImplementedError
