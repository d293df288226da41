"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Restatement of the four `torch_geometric.nn` operators the MS-HGNN hot path uses, following the
published behaviour of torch_geometric==2.5.0 (pinned by the reference at pyproject.toml:16 and
environment_files/requirements.txt:110; the package source is NOT vendored under /root/reference and
is not installable here).  Call sites these stand in for: hgnn_c2.py:88,100,106,113,131;
hgnn_k4.py:97,109,116,122,129,144; hgnn.py:34,41,44,55.

Published algorithm (PyG 2.5.0):
  * nn.dense.linear.Linear(in, out, bias): y = x W^T + b, W:[out,in]; in=-1 is lazy (materialised on the
    first forward); default reset = kaiming_uniform(a=sqrt(5)) on W (== U(+-1/sqrt(in))) and
    U(+-1/sqrt(in)) on b.
  * nn.dense.linear.HeteroDictLinear(in, out, types): one Linear per node type in `lins[type]`,
    forward maps each present key.
  * nn.conv.GraphConv(in, out, aggr): out_i = lin_rel(aggr_{j->i} x_src[j]) + lin_root(x_dst[i]);
    aggregation happens BEFORE lin_rel (bias added once per destination node, also for nodes with no
    in-edges); lin_root has no bias; 'mean' divides by max(in-degree, 1).
  * nn.conv.HeteroConv(convs, aggr='sum'): iterate relations in constructor order, skip relations absent
    from edge_index_dict, call conv(x_src, ei) if src==dst else conv((x_src, x_dst), ei); per destination
    type stack the results and sum over the relation axis.  Submodules are stored under the key
    '<src___rel___dst>' (nn.module_dict.ModuleDict.to_internal_key).
"""
import math
from typing import Dict, Tuple

import torch
from torch import nn


class Linear(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        if in_channels > 0:
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        else:
            self.weight = nn.parameter.UninitializedParameter()
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        if in_channels > 0:
            self.reset_parameters()

    def reset_parameters(self):
        if self.in_channels <= 0:
            return
        bound = 1.0 / math.sqrt(self.in_channels)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        if isinstance(self.weight, nn.parameter.UninitializedParameter):
            self.in_channels = x.shape[-1]
            self.weight.materialize((self.out_channels, self.in_channels))
            self.reset_parameters()
        return torch.nn.functional.linear(x, self.weight, self.bias)


class HeteroDictLinear(nn.Module):
    def __init__(self, in_channels, out_channels: int, types=None, **kwargs):
        super().__init__()
        if isinstance(in_channels, dict):
            types = list(in_channels.keys())
            ins = in_channels
        else:
            ins = {t: in_channels for t in types}
        self.lins = nn.ModuleDict({t: Linear(ins[t], out_channels, bias=True) for t in types})

    def reset_parameters(self):
        for lin in self.lins.values():
            lin.reset_parameters()

    def forward(self, x_dict):
        return {k: self.lins[k](x) for k, x in x_dict.items() if k in self.lins}


class GraphConv(nn.Module):
    def __init__(self, in_channels, out_channels: int, aggr: str = "add", bias: bool = True, **kwargs):
        super().__init__()
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        assert aggr in ("add", "sum", "mean")
        self.aggr = aggr
        self.lin_rel = Linear(in_channels[0], out_channels, bias=bias)
        self.lin_root = Linear(in_channels[1], out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None, size=None):
        if isinstance(x, torch.Tensor):
            x = (x, x)
        x_src, x_dst = x
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros(x_dst.shape[0], x_src.shape[1], dtype=x_src.dtype, device=x_src.device)
        agg = agg.index_add(0, dst, x_src.index_select(0, src))
        if self.aggr == "mean":
            deg = torch.zeros(x_dst.shape[0], dtype=x_src.dtype, device=x_src.device)
            deg = deg.index_add(0, dst, torch.ones_like(dst, dtype=x_src.dtype))
            agg = agg / deg.clamp(min=1).unsqueeze(-1)
        out = self.lin_rel(agg)
        if x_dst is not None:
            out = out + self.lin_root(x_dst)
        return out


def _internal_key(edge_type: Tuple[str, str, str]) -> str:
    return "<" + "___".join(edge_type) + ">"


class HeteroConv(nn.Module):
    def __init__(self, convs: Dict[Tuple[str, str, str], nn.Module], aggr: str = "sum"):
        super().__init__()
        assert aggr == "sum"
        self._edge_types = list(convs.keys())
        self.convs = nn.ModuleDict({_internal_key(k): v for k, v in convs.items()})
        self.aggr = aggr

    def reset_parameters(self):
        for c in self.convs.values():
            c.reset_parameters()

    def forward(self, x_dict, edge_index_dict):
        outs = {}
        for et in self._edge_types:
            if et not in edge_index_dict:
                continue
            src, _, dst = et
            if src not in x_dict or dst not in x_dict:
                continue
            conv = self.convs[_internal_key(et)]
            if src == dst:
                o = conv(x_dict[src], edge_index_dict[et])
            else:
                o = conv((x_dict[src], x_dict[dst]), edge_index_dict[et])
            outs.setdefault(dst, []).append(o)
        return {k: torch.stack(v, dim=0).sum(dim=0) for k, v in outs.items()}
