"""Operator surface of the hot path: the four `torch_geometric.nn` names the reference imports
(hgnn_c2.py:3) -- `Linear`, `HeteroDictLinear`, `GraphConv`, `HeteroConv` -- as parameter containers with
the same constructor signatures, attribute names and state_dict keys (PyG 2.5.0):

    encoder.lins.<type>.{weight,bias}
    convs.<l>.convs.<src___rel___dst>.lin_rel.{weight,bias}, .lin_root.weight
    decoder.{weight,bias}

so reference checkpoints load unchanged and `model.convs[i].convs[edge_type].lin_rel.weight`
(hgnn_c2.py:295-306) keeps working.  Their arithmetic does NOT live here: the model classes in models.py hand
all parameters to the fused HIP engine.  Calling one of these modules on its own raises, loudly -- there is
no eager / CPU implementation of the path in this package.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import nn

EdgeType = Tuple[str, str, str]

_STANDALONE = ("{} is a parameter container of the fused MI355X MS-HGNN engine; run it through "
               "GRF_HGNN_C2 / GRF_HGNN_K4 / GRF_HGNN (morphsym_hgnn_amd.models). There is no eager fallback.")


class Linear(nn.Module):
    """PyG `Linear(in_channels, out_channels, bias=True)`; `in_channels=-1` is lazy (materialised by the
    model on its first forward, mirroring gnnLightning.py:593-595)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        if in_channels > 0:
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        else:
            self.weight = nn.parameter.UninitializedParameter()
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def materialize(self, in_channels: int):
        if isinstance(self.weight, nn.parameter.UninitializedParameter):
            self.in_channels = in_channels
            self.weight.materialize((self.out_channels, in_channels))
            self.reset_parameters()

    def reset_parameters(self):
        if self.in_channels <= 0 or isinstance(self.weight, nn.parameter.UninitializedParameter):
            return
        bound = 1.0 / math.sqrt(self.in_channels)   # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        raise NotImplementedError(_STANDALONE.format("Linear"))


class HeteroDictLinear(nn.Module):
    def __init__(self, in_channels, out_channels: int, types=None, **kwargs):
        super().__init__()
        if isinstance(in_channels, dict):
            types = list(in_channels.keys())
            ins = in_channels
        else:
            ins = {t: in_channels for t in types}
        self.out_channels = out_channels
        self.lins = nn.ModuleDict({t: Linear(ins[t], out_channels, bias=True) for t in types})

    def reset_parameters(self):
        for lin in self.lins.values():
            lin.reset_parameters()

    def forward(self, x_dict):
        raise NotImplementedError(_STANDALONE.format("HeteroDictLinear"))


class GraphConv(nn.Module):
    def __init__(self, in_channels, out_channels: int, aggr: str = "add", bias: bool = True, **kwargs):
        super().__init__()
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        if aggr not in ("add", "sum", "mean"):
            raise ValueError(f"unsupported aggregation {aggr!r}")
        self.aggr = "add" if aggr == "sum" else aggr
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_rel = Linear(in_channels[0], out_channels, bias=bias)
        self.lin_root = Linear(in_channels[1], out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None, size=None):
        raise NotImplementedError(_STANDALONE.format("GraphConv"))


def internal_key(edge_type) -> str:
    """PyG ModuleDict.to_internal_key: tuple keys become '<a___b___c>'."""
    if isinstance(edge_type, tuple):
        return "<" + "___".join(edge_type) + ">"
    return edge_type


class RelationDict(nn.ModuleDict):
    """ModuleDict that accepts (src, rel, dst) tuples as keys, like torch_geometric.nn.module_dict.ModuleDict."""

    def __getitem__(self, key):
        return super().__getitem__(internal_key(key))

    def __setitem__(self, key, module):
        super().__setitem__(internal_key(key), module)

    def __contains__(self, key):
        return super().__contains__(internal_key(key))


class HeteroConv(nn.Module):
    def __init__(self, convs: Dict[EdgeType, nn.Module], aggr: str = "sum"):
        super().__init__()
        if aggr != "sum":
            raise ValueError("the MS-HGNN path uses HeteroConv(aggr='sum') only (hgnn_c2.py:113)")
        self.aggr = aggr
        self.edge_types = [tuple(k) for k in convs.keys()]
        self.convs = RelationDict()
        for k, v in convs.items():
            self.convs[tuple(k)] = v

    def reset_parameters(self):
        for c in self.convs.values():
            c.reset_parameters()

    def forward(self, x_dict, edge_index_dict):
        raise NotImplementedError(_STANDALONE.format("HeteroConv"))
