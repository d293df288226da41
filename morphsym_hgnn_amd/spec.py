"""Model specification of the MS-HGNN hot path: which nodes, relations, widths, symmetry masks and
parameters one model instance has.  Pure host logic (no GPU); the plan compiler (plan.py) lowers a
ModelSpec to the integer tables the HIP kernels interpret.

Reference anchors: hgnn_c2.py:10-131 (C2), hgnn_k4.py:10-144 (K4), hgnn.py:10-55 (MI-HGNN baseline).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .topology import EdgeType, RobotTopology

KINDS = ("c2", "k4", "mi", "k4_com", "c2_com", "s4_com")
COM_KINDS = ("k4_com", "c2_com", "s4_com")   # Solo centroidal-momentum variants: decoder on base nodes, T=1 (hgnn_*_com.py)


def rel_key(et: EdgeType) -> str:
    """State-dict key fragment PyG's ModuleDict uses for a relation triple: '<src___rel___dst>'."""
    return "<" + "___".join(et) + ">"


def relation_aggr(kind: str, et: EdgeType) -> str:
    """GraphConv aggregation per relation: 'mean' for center_bb (hgnn_c2.py:98-104) and gt/gs
    (hgnn_k4.py:107-119), 'add' otherwise."""
    if kind == "c2" and et[1] == "center_bb":
        return "mean"
    if kind in ("k4", "k4_com", "c2_com") and et[1] in ("gt", "gs"):   # hgnn_k4_com.py:93-103, hgnn_c2_com.py:72-83
        return "mean"
    return "add"


@dataclass
class ModelSpec:
    kind: str
    topology: RobotTopology
    hidden: int
    num_layers: int
    widths: Dict[str, int]                  # input feature width per node type
    regression: bool = True
    grf_dimension: int = 3
    group: Optional[dict] = None            # parsed group-operator YAML (None => all masks +1)
    num_timesteps: int = 150                # hgnn_c2.py:30 / hgnn_k4.py:29
    out_type: str = "foot"
    com_dimension: int = 6                   # decoder width per base node of the COM variants (hgnn.py:73,93)

    def __post_init__(self):
        if self.kind not in KINDS:
            raise ValueError(f"unknown model kind {self.kind!r}")
        for t in self.node_types:
            if t not in self.widths:
                raise ValueError(f"missing input width for node type {t!r}")
        if self.kind in COM_KINDS:
            self.out_type = "base"
            self.num_timesteps = 1
            nb = {"k4_com": 4, "c2_com": 2, "s4_com": 1}[self.kind]
            nn_ = self.topology.num_nodes
            if nn_.get("base") != nb or (self.kind != "s4_com" and "foot" in nn_):
                raise ValueError(f"{self.kind} model expects {nb} base nodes and no foot type, got {nn_}")
            if self.kind != "s4_com" and self.com_dimension != 6:
                raise ValueError("COM K4/C2 models decode 6 values (lin 3 + ang 3) per base node")
            if self.kind != "s4_com" and (nn_.get("joint") != 12 or self.widths["joint"] != 2):
                raise ValueError("COM K4/C2 models expect 12 joint nodes with 2 features (hgnn_k4_com.py:163-166)")
        if self.kind in ("c2", "k4"):
            nb = 2 if self.kind == "c2" else 4
            nn_ = self.topology.num_nodes
            if nn_.get("base") != nb or nn_.get("joint") != 12 or nn_.get("foot") != 4:
                raise ValueError(f"{self.kind} model expects {nb} base / 12 joint / 4 foot nodes "
                                 f"(hgnn_{self.kind}.py:30-36), got {nn_}")
            T = self.num_timesteps
            if self.widths["base"] != 6 * T:
                raise ValueError("base width must be 6*num_timesteps (hgnn_c2.py:246-251)")
            nv = self.joint_vars
            if self.widths["joint"] != nv * T:
                raise ValueError(f"joint width must be {nv}*num_timesteps (hgnn_c2.py:199 / hgnn_k4.py:206)")
            if self.masks_foot_inputs and self.widths["foot"] != 6 * T:
                raise ValueError("foot width must be 6*num_timesteps when foot inputs are masked")

    # ---- structure -------------------------------------------------------------------------
    @property
    def node_types(self) -> List[str]:
        return self.topology.node_types

    @property
    def edge_types(self) -> List[EdgeType]:
        return self.topology.edge_types

    @property
    def num_nodes(self) -> Dict[str, int]:
        return self.topology.num_nodes

    @property
    def joint_vars(self) -> int:
        if self.kind == "k4":
            return 2                                   # hgnn_k4.py:206
        return 3 if self.regression else 2             # hgnn_c2.py:37-40

    @property
    def masks_foot_inputs(self) -> bool:
        return self.kind == "k4" or (self.kind == "c2" and not self.regression)

    @property
    def has_base_transform(self) -> bool:
        return self.kind in ("c2", "k4", "k4_com", "c2_com")

    @property
    def residual(self) -> bool:
        return self.kind in ("c2", "k4", "k4_com", "c2_com")

    @property
    def out_channels(self) -> int:
        """out_channels_per_foot (hgnn_c2.py:124-129, hgnn_k4.py:139-143, hgnn.py:49-54); 6 per base for COM."""
        if self.kind in COM_KINDS:
            return self.com_dimension                  # num_dimensions_per_base, hgnn_k4_com.py:34,123
        if self.kind == "k4":
            return 1 if self.regression else 2
        if self.regression and self.grf_dimension == 1:
            return 1
        if self.regression and self.grf_dimension == 3:
            return 3
        return 2

    def live_types(self, layer: int) -> List[str]:
        """Node types whose output of message-passing layer `layer` can reach the decoder (the engine skips
        the others: e.g. base/joint outputs of the last layer are never read, hgnn_c2.py:176)."""
        live = {self.out_type}
        for l in range(self.num_layers - 2, layer - 1, -1):
            nxt = set(live)
            for s, _, d in self.edge_types:
                if d in live:
                    nxt.add(s)
            live = nxt
        return [t for t in self.node_types if t in live]

    def node_liveness(self):
        """Node-level liveness, the mirror of the plan compiler's (csrc/mshgnn_plan.hpp): returns (live, need) with
        live[l][t] = the nodes of type t (indices inside the type) whose output of layer l can reach the decoder, and need[l][t] = the nodes whose
        X_l is an input of a live node of layer l (need[0] = what the encoder has to compute; need[l] == live[l - 1] for l >= 1).  X_{l+1}[n] is needed
        iff n itself is live in layer l + 1 (root weight, residual) or has an edge into a node that is; the base_transform type is live as a whole.
        A1-C2 at 3 layers: the base nodes are four hops from the feet -- nothing of them is live, in the reference either (49 of its 84 parameter
        tensors get exact-zero gradients).  MSHGNN_PRUNE=0 (read like the plan compiler does): whole types, the liveness of rounds 1-3."""
        import os
        prune = os.environ.get("MSHGNN_PRUNE", "1") != "0"
        types, L = self.node_types, self.num_layers
        rels = [(s, d, self.topology.edges(et)) for et in self.edge_types for s, _, d in [et]]

        def widen(sel):
            for t in types:
                if prune and not (self.has_base_transform and t == "base"):
                    continue
                if sel[t]:
                    sel[t] = set(range(self.num_nodes[t]))

        def inputs_of(out):
            inp = {t: set(out[t]) for t in types}
            for s, d, edges in rels:
                for j, i in edges:
                    if i in out[d]:
                        inp[s].add(j)
                if not prune and out[d]:
                    inp[s] = set(range(self.num_nodes[s]))
            return inp

        live = [None] * L
        need = [None] * L
        cur = {t: (set(range(self.num_nodes[t])) if t == self.out_type else set()) for t in types}
        for l in range(L - 1, -1, -1):
            widen(cur)
            live[l] = {t: set(cur[t]) for t in types}
            need[l] = inputs_of(live[l])
            if l == 0:
                widen(need[0])
            cur = {t: set(need[l][t]) for t in types}
        for l in range(1, L):
            need[l] = {t: set(live[l - 1][t]) for t in types}
        srt = lambda dd: {t: sorted(v) for t, v in dd.items()}
        return [srt(x) for x in live], [srt(x) for x in need]

    def dead_parameters(self):
        """The state_dict names whose gradient is an exact zero at this depth (node_liveness): every use of the tensor sits on nodes that cannot reach
        the decoder.  The reference's autograd leaves them None / zero, the engine writes zeros; a data-parallel exchange may skip them."""
        live, need = self.node_liveness()
        dead = set()
        for t in self.node_types:
            if not need[0][t]:
                dead |= {f"encoder.lins.{t}.weight", f"encoder.lins.{t}.bias"}
        for l in range(self.num_layers):
            for et in self.edge_types:
                pre = f"convs.{l}.convs.{rel_key(et)}."
                if not live[l][et[2]]:                                   # root weight and bias act on the relation's destination type
                    dead |= {pre + "lin_root.weight", pre + "lin_rel.bias"}
                if not any(i in live[l][et[2]] for _, i in self.topology.edges(et)):
                    dead.add(pre + "lin_rel.weight")
        if self.has_base_transform and not any(live[l]["base"] for l in range(self.num_layers)):
            dead |= {"base_transform.0.weight", "base_transform.0.bias", "base_transform.2.weight", "base_transform.2.bias"}
        return dead

    def live_gradient_index(self):
        """int64 indices into the flat gradient buffer of every element that can be non-zero (every parameter not in dead_parameters), ascending."""
        import torch as _t
        dead = self.dead_parameters()
        parts = [_t.arange(off, off + n, dtype=_t.int64) for k, (off, n) in self.param_offsets().items() if k not in dead]
        return _t.cat(parts) if parts else _t.zeros(0, dtype=_t.int64)

    # ---- parameters ------------------------------------------------------------------------
    def param_shapes(self) -> "OrderedDict[str, Tuple[int, ...]]":
        """state_dict names -> shapes, in module registration order."""
        h = self.hidden
        d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        for t in self.node_types:
            d[f"encoder.lins.{t}.weight"] = (h, self.widths[t])
            d[f"encoder.lins.{t}.bias"] = (h,)
        for l in range(self.num_layers):
            for et in self.edge_types:
                p = f"convs.{l}.convs.{rel_key(et)}."
                d[p + "lin_rel.weight"] = (h, h)
                d[p + "lin_rel.bias"] = (h,)
                d[p + "lin_root.weight"] = (h, h)
        if self.has_base_transform:
            d["base_transform.0.weight"] = (h, h)
            d["base_transform.0.bias"] = (h,)
            d["base_transform.2.weight"] = (h, h)
            d["base_transform.2.bias"] = (h,)
        d["decoder.weight"] = (self.out_channels, h)
        d["decoder.bias"] = (self.out_channels,)
        return d

    def param_offsets(self) -> "OrderedDict[str, Tuple[int, int]]":
        """name -> (offset, numel) in the flat fp32 parameter / gradient buffer (16-element aligned)."""
        cached = self.__dict__.get("_param_offsets")      # the spec is immutable after construction; this is on every call path
        if cached is not None:
            return cached
        off = 0
        out: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        for k, s in self.param_shapes().items():
            n = 1
            for v in s:
                n *= v
            out[k] = (off, n)
            off += (n + 15) // 16 * 16
        self.__dict__["_param_offsets"] = out
        return out

    def num_params(self) -> int:
        n = 0
        for s in self.param_shapes().values():
            m = 1
            for v in s:
                m *= v
            n += m
        return n

    def flat_size(self) -> int:
        cached = self.__dict__.get("_flat_size")          # the spec is immutable after construction; this is on every call path
        if cached is None:
            offs = self.param_offsets()
            last = next(reversed(offs.values()))
            cached = self.__dict__["_flat_size"] = (last[0] + last[1] + 15) // 16 * 16
        return cached

    # ---- symmetry masks --------------------------------------------------------------------
    def symmetry_coefficients(self):
        """(joint[12], foot[12], base_lin[3nb], base_ang[3nb]) float64 -- hgnn_c2.py:44-83, hgnn_k4.py:37-95."""
        f64 = torch.float64
        one3 = torch.ones(3, dtype=f64)
        if self.kind in ("mi", "s4_com"):
            return None
        nb = self.num_nodes["base"]
        g = self.group
        if g is None:
            return (torch.ones(12, dtype=f64), torch.ones(12, dtype=f64),
                    torch.ones(3 * nb, dtype=f64), torch.ones(3 * nb, dtype=f64))
        if self.kind in ("k4_com", "c2_com"):   # hgnn_k4_com.py:37-80, hgnn_c2_com.py:37-68 (no foot space)
            def row3(key, i):
                return torch.tensor(g[key][i][:3], dtype=f64)
            j_gs, bl_gs, ba_gs = row3("reflection_Q_js", 0), row3("reflection_Q_bs_lin", 0), row3("reflection_Q_bs_ang", 0)
            if self.kind == "c2_com":
                return (torch.cat((one3, one3, j_gs, j_gs)), None, torch.cat((one3, bl_gs)), torch.cat((one3, ba_gs)))
            j_gt, bl_gt, ba_gt = row3("reflection_Q_js", 1), row3("reflection_Q_bs_lin", 1), row3("reflection_Q_bs_ang", 1)
            return (torch.cat((one3, j_gt, j_gs, j_gs * j_gt)), None,
                    torch.cat((one3, bl_gt, bl_gs, bl_gs * bl_gt)), torch.cat((one3, ba_gt, ba_gs, ba_gs * ba_gt)))

        def row(key, i):
            return torch.tensor(g[key][i][:3], dtype=f64)

        j_gs, f_gs = row("reflection_Q_js", 0), row("reflection_Q_fs", 0)
        bl_gs, ba_gs = row("reflection_Q_bs_lin", 0), row("reflection_Q_bs_ang", 0)
        if self.kind == "c2":
            return (torch.cat((one3, one3, j_gs, j_gs)), torch.cat((one3, one3, f_gs, f_gs)),
                    torch.cat((one3, bl_gs)), torch.cat((one3, ba_gs)))
        j_gt, f_gt = row("reflection_Q_js", 1), row("reflection_Q_fs", 1)
        bl_gt, ba_gt = row("reflection_Q_bs_lin", 1), row("reflection_Q_bs_ang", 1)
        return (torch.cat((one3, j_gt, j_gs, j_gs * j_gt)), torch.cat((one3, f_gt, f_gs, f_gs * f_gt)),
                torch.cat((one3, bl_gt, bl_gs, bl_gs * bl_gt)), torch.cat((one3, ba_gt, ba_gs, ba_gs * ba_gt)))

    def input_masks(self) -> Dict[str, torch.Tensor]:
        """+-1 mask [n_type, F_type] (float64) per node type; all-ones where apply_symmetry does nothing.
        Net effect of apply_symmetry / unpack_data / pack_data (hgnn_c2.py:191-284)."""
        masks = {t: torch.ones(self.num_nodes[t], self.widths[t], dtype=torch.float64) for t in self.node_types}
        if self.kind in ("mi", "s4_com"):
            return masks
        if self.kind in ("k4_com", "c2_com"):    # only the joints are masked (hgnn_k4_com.py:159-168)
            cj = self.symmetry_coefficients()[0]
            masks["joint"] = cj.view(12, 1).expand(12, self.widths["joint"]).clone()
            return masks
        T = self.num_timesteps
        cj, cf, cbl, cba = self.symmetry_coefficients()
        masks["joint"] = cj.view(12, 1).expand(12, self.widths["joint"]).clone()
        nb = self.num_nodes["base"]
        lin = cbl.view(nb, 3, 1).expand(nb, 3, T).reshape(nb, 3 * T)
        ang = cba.view(nb, 3, 1).expand(nb, 3, T).reshape(nb, 3 * T)
        masks["base"] = torch.cat((lin, ang), dim=1)
        if self.masks_foot_inputs:
            masks["foot"] = cf.view(4, 1, 3, 1).expand(4, 2, 3, T).reshape(4, 6 * T).clone()
        return masks

    def output_mask(self) -> torch.Tensor:
        """[n_out, out_channels] +-1 mask on the decoder output: feet_linear_weights for C2 3-D GRF
        regression (hgnn_c2.py:179-189), ones otherwise."""
        n = self.num_nodes[self.out_type]
        m = torch.ones(n, self.out_channels, dtype=torch.float64)
        if self.kind == "c2" and self.regression and self.grf_dimension == 3:
            _, cf, _, _ = self.symmetry_coefficients()
            m = cf.view(n, 3).clone()
        if self.kind in ("k4_com", "c2_com"):   # morphological_symmetry_decoder: [lin(3) | ang(3)] per base node
            _, _, cbl, cba = self.symmetry_coefficients()
            m = torch.cat((cbl.view(n, 3), cba.view(n, 3)), dim=1)
        return m

    @property
    def output_is_window_major(self) -> bool:
        """True when forward returns [B, 4*out] (C2 3-D regression, ms_foot_decoder), else [B*4, out]."""
        return self.kind == "c2" and self.regression and self.grf_dimension == 3
