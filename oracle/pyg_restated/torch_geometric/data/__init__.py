"""Container half of torch_geometric.data (2.5.0) that the reference's dataset classes touch: HeteroData as a store of
per-node-type / per-edge-type attribute bags plus graph-level attributes.  No arithmetic."""


class _Store:
    pass


class HeteroData:
    def __init__(self):
        object.__setattr__(self, "_stores", {})

    def __getitem__(self, key):
        key = tuple(key) if isinstance(key, (tuple, list)) else key
        return self._stores.setdefault(key, _Store())

    @property
    def x_dict(self):
        return {k: v.x for k, v in self._stores.items() if isinstance(k, str) and hasattr(v, "x")}

    @property
    def edge_index_dict(self):
        return {k: v.edge_index for k, v in self._stores.items() if isinstance(k, tuple) and hasattr(v, "edge_index")}


class Data:
    pass


class Dataset:
    def __init__(self, *a, **k):
        pass
