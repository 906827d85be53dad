"""`torch_geometric.data` stand-in: names only. TEST INFRASTRUCTURE ONLY."""


class Data:  # pragma: no cover
    pass


class Dataset:  # pragma: no cover
    pass
