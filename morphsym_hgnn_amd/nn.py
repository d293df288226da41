"""Operator surface of the hot path: the four `torch_geometric.nn` names the reference imports
(hgnn_c2.py:3) -- `Linear`, `HeteroDictLinear`, `GraphConv`, `HeteroConv` -- as parameter containers with
the same constructor signatures, attribute names and state_dict keys (PyG 2.5.0):

    encoder.lins.<type>.{weight,bias}
    convs.<l>.convs.<src___rel___dst>.lin_rel.{weight,bias}, .lin_root.weight
    decoder.{weight,bias}

so reference checkpoints load unchanged and `model.convs[i].convs[edge_type].lin_rel.weight`
(hgnn_c2.py:295-306) keeps working.  The model classes in models.py hand all parameters to the fused HIP engine and
never call these forwards.  Called on their own (a maintainer who swaps only the PyG import) they run PyG 2.5.0's
semantics -- `forward(x)`, `forward(x_dict)`, `forward(x | (x_src, x_dst), edge_index)`, `forward(x_dict,
edge_index_dict)` -- on the stand-alone HIP operators of ops.py (fp32 MFMA, autograd), for any graph and width.  There
is no eager / CPU implementation: tensors off the HIP device raise.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import nn

EdgeType = Tuple[str, str, str]

from . import ops


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
        """y = x W^T + b; a lazy weight (`in_channels=-1`) is materialised from x on the first call, like PyG's."""
        self.materialize(x.shape[-1])
        return ops.linear(x, self.weight, self.bias)


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
        """One Linear per node type present in x_dict (PyG HeteroDictLinear.forward)."""
        return {k: self.lins[k](x) for k, x in x_dict.items() if k in self.lins}


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
        """out_i = lin_rel(aggr_{j->i} x_src[j]) + lin_root(x_dst[i]); x is a tensor or an (x_src, x_dst) pair."""
        if edge_weight is not None:
            raise NotImplementedError("GraphConv(edge_weight=...) is not on the MS-HGNN path (hgnn_c2.py never passes one)")
        return ops.graph_conv(x, edge_index, self.lin_rel.weight, self.lin_rel.bias, self.lin_root.weight, self.aggr)


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
        """PyG HeteroConv.forward: relations in constructor order, those without edges or features skipped, results summed per
        destination type (`torch.stack(xs).sum(0)`: left to right)."""
        outs: Dict[str, list] = {}
        for et in self.edge_types:
            if et not in edge_index_dict:
                continue
            src, _, dst = et
            if src not in x_dict or dst not in x_dict:
                continue
            outs.setdefault(dst, []).append(self.convs[et]((x_dict[src], x_dict[dst]), edge_index_dict[et]))
        res = {}
        for dst, xs in outs.items():
            acc = xs[0]
            for t in xs[1:]:
                acc = acc + t
            res[dst] = acc
        return res
