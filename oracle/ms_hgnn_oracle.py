"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (morphsym_hgnn_amd/*);
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.

CPU restatement (float64, plain torch tensor ops, no torch_geometric) of the reference's MS-HGNN hot
path: ``GRF_HGNN_C2.forward`` (src/ms_hgnn/lightning_py/hgnn_c2.py:133-182), ``GRF_HGNN_K4.forward``
(hgnn_k4.py:146-196) and the MI-HGNN baseline ``GRF_HGNN.forward`` (hgnn.py:57-62), plus the MSE loss of
the Lightning wrapper (gnnLightning.py:633-639, 680-695).  Gradients come from torch autograd over this
restatement, exactly as the reference obtains them.

PARITY PIN STATUS: the reference's own tests hold NO golden vector for this path (SURVEY.md section 4 / 8c:
tests/ never instantiates a C2/K4 model, the two MI-HGNN checkpoints are in .MISSING_LARGE_BLOBS) => by the
reference's tests alone this oracle is "parity unpinned".  It is pinned instead by outputs of the reference
itself run in the build container: oracle/gen_golden.py imports hgnn_c2.py / hgnn_k4.py / hgnn.py from
/root/reference by file path (over oracle/pyg_restated, a restatement of the four torch_geometric==2.5.0
operators the package uses -- PyG itself is absent and un-installable here), checks this file against them
to <=1e-12 on outputs, loss and every gradient, and commits the vectors under tests/golden/.  The exact
group-equivariance identity f(g.x) == g.f(x) (tests/test_oracle.py) pins masks + weight sharing + topology
independently of any restated code.

Calling convention = the reference's: ``x_dict[type]`` is [B*n_type, F_type] graph-major,
``edge_index_dict[(src, rel, dst)]`` is the PyG-batched LongTensor [2, B*E]; parameters are passed as a
dict keyed by the reference's state_dict names
  encoder.lins.<type>.{weight,bias}
  convs.<l>.convs.<src___rel___dst>.lin_rel.{weight,bias} / .lin_root.weight
  base_transform.{0,2}.{weight,bias}          (C2 / K4 only)
  decoder.{weight,bias}
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

EdgeType = Tuple[str, str, str]


@dataclass
class OracleConfig:
    kind: str                      # 'c2' | 'k4' | 'mi' | 'k4_com' | 'c2_com' | 's4_com'
    num_layers: int
    edge_types: Sequence[EdgeType]  # relation order of data_metadata[1]
    regression: bool = True
    grf_dimension: int = 3         # c2 / mi only (hgnn_c2.py:124-129)
    num_timesteps: int = 150       # hard-coded in the reference (hgnn_c2.py:30, hgnn_k4.py:29)
    group: Optional[dict] = None   # parsed group-operator YAML, or None for symmetry_mode=None

    @property
    def out_channels_per_foot(self) -> int:
        if self.kind.endswith("_com"):
            return 6                                    # num_dimensions_per_base, hgnn_k4_com.py:34
        if self.kind == "k4":
            return 1 if self.regression else 2          # hgnn_k4.py:139-143
        if self.regression and self.grf_dimension == 1:  # hgnn_c2.py:124-129 / hgnn.py:49-54
            return 1
        if self.regression and self.grf_dimension == 3:
            return 3
        return 2


def rel_key(et: EdgeType) -> str:
    """PyG ModuleDict internal key for a relation triple."""
    return "<" + "___".join(et) + ">"


# ----------------------------------------------------------------------------------------------
# Symmetry coefficients (hgnn_c2.py:44-85, hgnn_k4.py:37-95)
# ----------------------------------------------------------------------------------------------
def symmetry_coefficients(cfg: OracleConfig):
    """Returns (joint[12], foot[12], base_lin[3*nb], base_ang[3*nb]) as float64 tensors."""
    f64 = torch.float64
    one3 = torch.ones(3, dtype=f64)
    if cfg.kind in ("mi", "s4_com"):
        return None
    g = cfg.group
    if cfg.kind in ("k4_com", "c2_com"):    # hgnn_k4_com.py:37-80 / hgnn_c2_com.py:37-68: joint + base lin/ang, no foot space
        nb = 4 if cfg.kind == "k4_com" else 2
        if g is None:
            return (torch.ones(12, dtype=f64), None, torch.ones(3 * nb, dtype=f64), torch.ones(3 * nb, dtype=f64))
        j_gs = torch.tensor(g["reflection_Q_js"][0][:3], dtype=f64)
        bl_gs = torch.tensor(g["reflection_Q_bs_lin"][0][:3], dtype=f64)
        ba_gs = torch.tensor(g["reflection_Q_bs_ang"][0][:3], dtype=f64)
        if cfg.kind == "c2_com":
            return (torch.cat((one3, one3, j_gs, j_gs)), None, torch.cat((one3, bl_gs)), torch.cat((one3, ba_gs)))
        j_gt = torch.tensor(g["reflection_Q_js"][1][:3], dtype=f64)
        bl_gt = torch.tensor(g["reflection_Q_bs_lin"][1][:3], dtype=f64)
        ba_gt = torch.tensor(g["reflection_Q_bs_ang"][1][:3], dtype=f64)
        return (torch.cat((one3, j_gt, j_gs, j_gs * j_gt)), None,
                torch.cat((one3, bl_gt, bl_gs, bl_gs * bl_gt)), torch.cat((one3, ba_gt, ba_gs, ba_gs * ba_gt)))
    if g is None:
        nb = 2 if cfg.kind == "c2" else 4
        return (torch.ones(12, dtype=f64), torch.ones(12, dtype=f64),
                torch.ones(3 * nb, dtype=f64), torch.ones(3 * nb, dtype=f64))
    j_gs = torch.tensor(g["reflection_Q_js"][0][:3], dtype=f64)
    f_gs = torch.tensor(g["reflection_Q_fs"][0][:3], dtype=f64)
    bl_gs = torch.tensor(g["reflection_Q_bs_lin"][0][:3], dtype=f64)
    ba_gs = torch.tensor(g["reflection_Q_bs_ang"][0][:3], dtype=f64)
    if cfg.kind == "c2":
        # joints = [FL, RL, FR, RR]: cat(e, e, gs, gs)   hgnn_c2.py:73,76,80,82
        return (torch.cat((one3, one3, j_gs, j_gs)), torch.cat((one3, one3, f_gs, f_gs)),
                torch.cat((one3, bl_gs)), torch.cat((one3, ba_gs)))
    j_gt = torch.tensor(g["reflection_Q_js"][1][:3], dtype=f64)
    f_gt = torch.tensor(g["reflection_Q_fs"][1][:3], dtype=f64)
    bl_gt = torch.tensor(g["reflection_Q_bs_lin"][1][:3], dtype=f64)
    ba_gt = torch.tensor(g["reflection_Q_bs_ang"][1][:3], dtype=f64)
    # cat(e, gt, gs, gr) with gr = gs*gt   hgnn_k4.py:82-91
    return (torch.cat((one3, j_gt, j_gs, j_gs * j_gt)), torch.cat((one3, f_gt, f_gs, f_gs * f_gt)),
            torch.cat((one3, bl_gt, bl_gs, bl_gs * bl_gt)), torch.cat((one3, ba_gt, ba_gs, ba_gs * ba_gt)))


def _axis_major_mask(coeff: torch.Tensor, num_nodes: int, T: int) -> torch.Tensor:
    """Mask for a node feature laid out [var0: x(T) y(T) z(T), var1: x(T) y(T) z(T)] where both
    variables share `coeff[node*3+axis]` -- the net effect of unpack_data -> multiply -> pack_data
    (hgnn_c2.py:233-284): element (node, v*3T + a*T + t) is scaled by coeff[node*3 + a]."""
    c = coeff.view(num_nodes, 1, 3, 1).expand(num_nodes, 2, 3, T)
    return c.reshape(num_nodes, 6 * T)


def input_masks(cfg: OracleConfig, num_nodes: Dict[str, int], widths: Dict[str, int]) -> Dict[str, torch.Tensor]:
    """+-1 mask per (node, feature) that `apply_symmetry` applies to each node type
    (hgnn_c2.py:191-231; hgnn_k4.py:198-236).  Types that are not masked are absent."""
    if cfg.kind in ("mi", "s4_com"):
        return {}
    if cfg.kind in ("k4_com", "c2_com"):   # apply_symmetry masks the joints only (hgnn_k4_com.py:159-168)
        cj = symmetry_coefficients(cfg)[0]
        return {"joint": cj.view(12, 1).expand(12, widths["joint"]).clone()}
    T = cfg.num_timesteps
    cj, cf, cbl, cba = symmetry_coefficients(cfg)
    masks = {}
    # joint: view(-1, 12, T, nvars) * w_j.view(1,-1,1,1)  => constant per joint node
    masks["joint"] = cj.view(12, 1).expand(12, widths["joint"]).clone()
    # base: lin = first 3T features, ang = last 3T, each with its own coefficient vector
    nb = num_nodes["base"]
    lin = cbl.view(nb, 3, 1).expand(nb, 3, T).reshape(nb, 3 * T)
    ang = cba.view(nb, 3, 1).expand(nb, 3, T).reshape(nb, 3 * T)
    masks["base"] = torch.cat((lin, ang), dim=1)
    # foot inputs: masked for K4 always (hgnn_k4.py:213-224), for C2 only when classifying (hgnn_c2.py:206)
    if cfg.kind == "k4" or not cfg.regression:
        masks["foot"] = _axis_major_mask(cf, 4, T)
    return masks


def apply_symmetry(cfg: OracleConfig, x_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Restatement of apply_symmetry as the elementwise +-1 masks it amounts to."""
    if cfg.kind in ("mi", "s4_com"):
        return dict(x_dict)
    nj, nf = 12, 4
    nb = 2 if cfg.kind in ("c2", "c2_com") else 4
    widths = {k: v.shape[1] for k, v in x_dict.items()}
    T = 1 if cfg.kind.endswith("_com") else cfg.num_timesteps
    if not cfg.kind.endswith("_com") and widths["base"] != 6 * T:
        raise RuntimeError("base feature width must be 6*num_timesteps (hgnn_c2.py:246-251)")
    masks = input_masks(cfg, {"base": nb, "joint": nj, "foot": nf}, widths)
    out = dict(x_dict)
    for t, m in masks.items():
        x = x_dict[t]
        n = m.shape[0]
        out[t] = (x.view(-1, n, x.shape[1]) * m.to(x.dtype).unsqueeze(0)).reshape(x.shape)
    return out


# ----------------------------------------------------------------------------------------------
# GraphConv / HeteroConv restated (torch_geometric 2.5.0 semantics, see oracle/pyg_restated)
# ----------------------------------------------------------------------------------------------
def graph_conv(x_src, x_dst, edge_index, w_rel, b_rel, w_root, aggr: str):
    src, dst = edge_index[0], edge_index[1]
    agg = torch.zeros(x_dst.shape[0], x_src.shape[1], dtype=x_src.dtype)
    agg = agg.index_add(0, dst, x_src.index_select(0, src))
    if aggr == "mean":
        deg = torch.zeros(x_dst.shape[0], dtype=x_src.dtype).index_add(0, dst, torch.ones(dst.shape[0], dtype=x_src.dtype))
        agg = agg / deg.clamp(min=1).unsqueeze(-1)
    return agg @ w_rel.t() + b_rel + x_dst @ w_root.t()


def relation_aggr(cfg: OracleConfig, et: EdgeType) -> str:
    rel = et[1]
    if cfg.kind == "c2" and rel == "center_bb":   # hgnn_c2.py:98-104
        return "mean"
    if cfg.kind in ("k4", "k4_com", "c2_com") and rel in ("gt", "gs"):  # hgnn_k4.py:107-119, hgnn_k4_com.py:93-103
        return "mean"
    return "add"


def hetero_conv(cfg: OracleConfig, params, layer: int, x_dict, edge_index_dict):
    outs: Dict[str, List[torch.Tensor]] = {}
    for et in cfg.edge_types:
        et = tuple(et)
        if et not in edge_index_dict:
            continue
        s, _, d = et
        p = f"convs.{layer}.convs.{rel_key(et)}."
        o = graph_conv(x_dict[s], x_dict[d], edge_index_dict[et],
                       params[p + "lin_rel.weight"], params[p + "lin_rel.bias"],
                       params[p + "lin_root.weight"], relation_aggr(cfg, et))
        outs.setdefault(d, []).append(o)
    return {k: torch.stack(v, dim=0).sum(dim=0) for k, v in outs.items()}


# ----------------------------------------------------------------------------------------------
# Whole-model forward
# ----------------------------------------------------------------------------------------------
def forward(cfg: OracleConfig, params: Dict[str, torch.Tensor], x_dict, edge_index_dict,
            return_hidden: bool = False, relu_fn=None):
    """hgnn_c2.py:133-182 / hgnn_k4.py:146-196 / hgnn.py:57-62.

    relu_fn(key, h) (tests only) replaces torch.relu at the site `key` = ("enc", type) | ("layer", l, type) | ("t1", l): the GPU
    parity harness passes h * (the engine's own relu decisions), so that a pre-activation within rounding error of zero, whose
    decision may legitimately differ from the exact one, does not turn into a spurious gradient mismatch (tests/helpers.py)."""
    relu = (lambda key, h: torch.relu(h)) if relu_fn is None else relu_fn
    x = apply_symmetry(cfg, x_dict)                                    # :143
    x = {k: relu(("enc", k), v @ params[f"encoder.lins.{k}.weight"].t() + params[f"encoder.lins.{k}.bias"])
         for k, v in x.items()}                                         # :146-147
    hidden = [x]
    for layer in range(cfg.num_layers):                                 # :150
        h = hetero_conv(cfg, params, layer, x, edge_index_dict)         # :152
        if cfg.kind in ("mi", "s4_com"):
            x = {k: relu(("layer", layer, k), v) for k, v in h.items()}     # hgnn.py:60-61 / hgnn_s4_com.py:67-69
        else:
            new = {}
            for k, v in h.items():
                if k == "base":                                          # :155-158 (base_transform shared across layers)
                    t1 = relu(("t1", layer), v @ params["base_transform.0.weight"].t() + params["base_transform.0.bias"])
                    new[k] = t1 @ params["base_transform.2.weight"].t() + params["base_transform.2.bias"]
                else:
                    new[k] = relu(("layer", layer, k), v)
            x = {k: new[k] + x[k] if (k in x and x[k].shape == new[k].shape) else new[k] for k in new}  # :161-166
        hidden.append(x)
    if cfg.kind.endswith("_com"):
        out = x["base"] @ params["decoder.weight"].t() + params["decoder.bias"]   # hgnn_k4_com.py:154 (decoder on base nodes)
        if cfg.kind != "s4_com":                                                   # morphological_symmetry_decoder :157-165
            _, _, cbl, cba = symmetry_coefficients(cfg)
            nb = cbl.numel() // 3
            o = out.reshape(-1, nb, 6)
            out = torch.cat((o[:, :, :3] * cbl.view(1, nb, 3).to(o.dtype), o[:, :, 3:] * cba.view(1, nb, 3).to(o.dtype)), dim=-1)
        if return_hidden:
            return out, hidden
        return out
    out = x["foot"] @ params["decoder.weight"].t() + params["decoder.bias"]   # :176
    if cfg.kind == "c2" and cfg.regression and cfg.grf_dimension == 3:        # :179-180, 184-189
        _, cf, _, _ = symmetry_coefficients(cfg)
        out = out.view(-1, 4, 3).flatten(start_dim=1) * cf.to(out.dtype)
    if return_hidden:
        return out, hidden
    return out


def wrapper_outputs(cfg: OracleConfig, out_raw: torch.Tensor, y: torch.Tensor, batch_size: int):
    """step_helper_function reshape contract (gnnLightning.py:680-695 regression; :498-513 classification,
    where labels are one {0,1} per foot: [B, 4])."""
    if cfg.kind.endswith("_com"):   # COM wrappers compare the [B, n_base*6] prediction with same-shape labels
        return y.reshape(batch_size, -1), out_raw.reshape(batch_size, -1)
    w = cfg.out_channels_per_foot * 4
    if out_raw.numel() != batch_size * w:      # not a quadruped (synthetic many-limb robot): the MSE is over the flattened tensors either way
        w = out_raw.numel() // batch_size
    return y.reshape(batch_size, w if cfg.regression else 4), out_raw.squeeze().reshape(batch_size, w)


def mse_loss(y: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """MeanSquaredError of the flattened batch (gnnLightning.py:633-639): mean((y_pred - y)^2)."""
    return ((y_pred.flatten() - y.flatten()) ** 2).mean()


def cross_entropy_loss(y: torch.Tensor, y_pred: torch.Tensor, batch_size: int) -> torch.Tensor:
    """CE of per-foot 2-class logits (gnnLightning.py:132-141, customMetrics.py:6-24): logits [B,8] ->
    [B*4, 2], targets y.long().flatten(); mean over B*4."""
    logits = y_pred.reshape(batch_size * 4, 2)
    return torch.nn.functional.cross_entropy(logits, y.long().flatten())


def step(cfg: OracleConfig, params: Dict[str, torch.Tensor], x_dict, edge_index_dict, y, batch_size: int):
    """One fwd + loss + bwd of the hot path.  Returns (out_raw, loss, grads dict)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = forward(cfg, leaves, {k: v.clone() for k, v in x_dict.items()}, edge_index_dict)
    yy, yp = wrapper_outputs(cfg, out, y, batch_size)
    loss = mse_loss(yy, yp) if cfg.regression else cross_entropy_loss(yy, yp, batch_size)
    loss.backward()
    return out.detach(), loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
