"""`torch_geometric.nn` stand-in: MetaLayer with the published PyG forward semantics
(row, col = edge_index; edge model sees x[row], x[col], edge_attr, u, batch[row];
then node model, then global model).  TEST INFRASTRUCTURE ONLY."""
import torch


class MetaLayer(torch.nn.Module):
    def __init__(self, edge_model=None, node_model=None, global_model=None):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model
        self.global_model = global_model

    def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
        row, col = edge_index[0], edge_index[1]
        if self.edge_model is not None:
            edge_attr = self.edge_model(x[row], x[col], edge_attr, u,
                                        batch if batch is None else batch[row])
        if self.node_model is not None:
            x = self.node_model(x, edge_index, edge_attr, u, batch)
        if self.global_model is not None:
            u = self.global_model(x, edge_index, edge_attr, u, batch)
        return x, edge_attr, u


class MessagePassing(torch.nn.Module):  # name only (UtilsGNN.py:41 imports it, never uses it)
    pass
