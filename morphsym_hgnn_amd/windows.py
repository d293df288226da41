"""On-device window assembly: a sequence's raw time series live on the GPU and a minibatch of time windows is gathered
straight into the engine's input layout by `mshgnn_assemble_windows` (include/mshgnn.h).

This is the device counterpart of the reference's per-window Python path (relative to /root/reference/src/ms_hgnn):
datasets_py/quadSDKDataset_Morph.py:99-175 (`load_data_sorted_c2`), :304-369 (`get_helper_heterogeneous_gnn_c2`),
:444-489 (`load_data_at_dataset_seq[_3d]`), datasets_py/flexibleDataset.py:340-400, 563-596 and PyG's collate -- i.e.
`DataLoader(dataset, batch_size=B)` -> `batch.x_dict`, `batch.y`, `batch.r_o` for B window indices at once.

A `WindowRecipe` says, per node type, which columns of which raw series become the T-long runs of a node's feature row
(variable-major, axis-major: the reference's `flatten('F')` layout); types without variables are all-ones of width 1
(flexibleDataset.py:183-190).  No CPU implementation: without a HIP device `SequenceStore` raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from . import engine as eng


@dataclass
class WindowRecipe:
    node_types: List[str]
    num_nodes: Dict[str, int]
    history: int                                            # T
    # per type: list of variables; a variable = (series name, columns[node][axis])
    variables: Dict[str, List[Tuple[str, List[List[int]]]]]
    label_series: Optional[str] = None
    label_cols: List[int] = field(default_factory=list)
    label_rotate: bool = False                              # 3-D world-frame labels -> body frame (needs quat_series)
    quat_series: Optional[str] = None
    normalize: bool = False                                 # per-window standardisation (flexibleDataset.py:390-396)

    def width(self, t: str) -> int:
        w = sum(len(cols[0]) * self.history for _, cols in self.variables.get(t, []))
        return w if w > 0 else 1

    def series(self) -> List[str]:
        names = []
        for t in self.node_types:
            for s, _ in self.variables.get(t, []):
                if s not in names:
                    names.append(s)
        for s in (self.label_series, self.quat_series):
            if s is not None and s not in names:
                names.append(s)
        return names


def quadsdk_a1_c2_recipe(joint_perm: Sequence[int], foot_perm: Sequence[int], history: int = 150, grf_dimension: int = 3,
                         body_frame_labels: bool = False, normalize: bool = False, n_base: int = 2) -> WindowRecipe:
    """The A1 / Quad-SDK C2 dataset (BASELINE config): base = tiled IMU (lin acc, ang vel), joint = (q, qd, tau) in graph
    order, feet = ones; labels = GRFs of the window's last step in foot order (quadSDKDataset_Morph.py:99-160, 304-369)."""
    if grf_dimension == 3:
        lab = [int(3 * i + k) for i in foot_perm for k in range(3)]
    elif grf_dimension == 1:
        lab = [int(3 * i + 2) for i in foot_perm]
    else:
        raise ValueError("grf_dimension must be 1 or 3")
    if body_frame_labels and grf_dimension != 3:
        # the reference rotates the 3-D GRFs first and then keeps z; on device that is the rotated z, which needs all three
        raise ValueError("body-frame labels need grf_dimension == 3")
    return WindowRecipe(
        node_types=["base", "joint", "foot"], num_nodes={"base": n_base, "joint": len(joint_perm), "foot": len(foot_perm)},
        history=history,
        variables={"base": [("imu_acc", [[0, 1, 2]] * n_base), ("imu_omega", [[0, 1, 2]] * n_base)],
                   "joint": [(s, [[int(j)] for j in joint_perm]) for s in ("q", "qd", "tau")], "foot": []},
        label_series="F", label_cols=lab, label_rotate=body_frame_labels, quat_series="r_o", normalize=normalize)


def minicheetah_k4_recipe(joint_perm: Sequence[int], foot_perm: Sequence[int], history: int = 150, normalize: bool = False,
                          n_base: int = 4) -> WindowRecipe:
    """The MiniCheetah (LinTzuYaun) contact dataset on the K4 graph: base = IMU tiled to the 4 base nodes, joint = (q, qd) in
    graph order, foot = (p, v) 3-D in foot order; labels = contact flags of the window's last step in foot order
    (LinTzuYaunDataset.py:65-88, LinTzuYaunDataset_Morph.py:251-347, 555-625)."""
    fcols = [[int(3 * i + k) for k in range(3)] for i in foot_perm]
    return WindowRecipe(
        node_types=["base", "joint", "foot"], num_nodes={"base": n_base, "joint": len(joint_perm), "foot": len(foot_perm)}, history=history,
        variables={"base": [("imu_acc", [[0, 1, 2]] * n_base), ("imu_omega", [[0, 1, 2]] * n_base)],
                   "joint": [(s, [[int(j)] for j in joint_perm]) for s in ("q", "qd")],
                   "foot": [("p", fcols), ("v", fcols)]},
        label_series="contacts", label_cols=[int(i) for i in foot_perm], normalize=normalize)


def solo_com_recipe(kind: str, joint_perm: Sequence[int], history: int = 1) -> WindowRecipe:
    """The Solo-12 centroidal-momentum dataset (BASELINE configs[3] data format) on the K4 / C2 / S4 graphs: joint = (q, qd) in graph
    order, base = the (all-zero) IMU series tiled to the base nodes, labels = the 6-D base velocity of the window's last row, once per
    base node as [lin(3) | ang(3)] (soloDataset.py:235-300, 382-400, 546-717).  Series: `solo_com_arrays(X, Y)`."""
    nb = {"k4_com": 4, "c2_com": 2, "s4_com": 1}[kind]
    return WindowRecipe(
        node_types=["base", "joint"], num_nodes={"base": nb, "joint": len(joint_perm)}, history=history,
        variables={"base": [("base_lin", [[0, 1, 2]] * nb), ("base_ang", [[0, 1, 2]] * nb)],
                   "joint": [(s, [[int(j)] for j in joint_perm]) for s in ("q", "qd")]},
        label_series="Y", label_cols=[0, 1, 2, 3, 4, 5] * nb)


def solo_com_arrays(X: np.ndarray, Y: np.ndarray) -> Dict[str, np.ndarray]:
    """The raw series `solo_com_recipe` names, from the dataset's X [N, 24] (q | qd) and Y [N, 6] (soloDataset.py:382-400: the base
    IMU inputs of this task are zeros)."""
    X, Y = np.asarray(X), np.asarray(Y)
    z = np.zeros((X.shape[0], 3), dtype=np.float32)
    return {"q": X[:, :12], "qd": X[:, 12:], "base_lin": z, "base_ang": z, "Y": Y}


class SequenceStore:
    """The raw series of one recorded sequence on the GPU + the recipe that turns window indices into engine inputs."""

    def __init__(self, arrays: Dict[str, np.ndarray], recipe: WindowRecipe, dtype: str = "bf16", device=None, fast: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("window assembly runs on a HIP device; there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.lib = eng.load_library()
        self.recipe = recipe
        self.dtype = dtype
        self.torch_dtype = torch.bfloat16 if dtype == "bf16" else torch.float32      # the split plan ("x3") takes fp32 inputs
        self.names = recipe.series()
        self.series = []
        n_rows = None
        self.series16 = []       # bf16 copies for the fused gather of Engine.step_mse_series (made on first use)
        for s in self.names:
            a = torch.as_tensor(np.asarray(arrays[s])).to(self.device, torch.float32)
            a = a.reshape(a.shape[0], -1).t()                       # column-major: one window of one column is contiguous
            n_rows = a.shape[1] if n_rows is None else min(n_rows, a.shape[1])
            pad = torch.zeros(a.shape[0], a.shape[1] + 8, dtype=torch.float32, device=self.device)     # 8 elements of slack behind every column
            pad[:, :a.shape[1]] = a                                 # (a 16-byte load of the fused gather may run past a window's last step)
            self.series.append(pad)
        self.n_rows = int(n_rows)
        if self.n_rows < recipe.history:
            raise ValueError("sequence shorter than one window")
        sidx = {s: i for i, s in enumerate(self.names)}
        runs, rows = [], []
        for ti, t in enumerate(recipe.node_types):
            vars_ = recipe.variables.get(t, [])
            if not vars_:
                for n in range(recipe.num_nodes[t]):
                    rows.append([len(runs), len(runs) + 1]); runs.append([ti, n, 0, -1, 1])
                continue
            for n in range(recipe.num_nodes[t]):
                f = 0
                r_begin = len(runs)
                for s, cols in vars_:
                    ncol = self.series[sidx[s]].shape[0]
                    for c in cols[n]:
                        if not 0 <= c < min(ncol, 256):
                            raise ValueError(f"column {c} of series '{s}' out of range")
                        runs.append([ti, n, f, (sidx[s] << 8) | c, recipe.history])
                        f += recipe.history
                rows.append([r_begin, len(runs)])
        self.runs = torch.tensor(runs, dtype=torch.int32, device=self.device)
        self.rows = torch.tensor(rows, dtype=torch.int32, device=self.device)
        self.label_cols = torch.tensor(recipe.label_cols or [0], dtype=torch.int32, device=self.device)
        d = eng.MshgnnWindowDesc()
        d.n_types = len(recipe.node_types); d.dtype = {"f32": 0, "bf16": 1, "x3": 2}[dtype]; d.history = recipe.history
        d.normalize = int(recipe.normalize)
        for i, t in enumerate(recipe.node_types):
            d.type_nodes[i] = recipe.num_nodes[t]; d.type_width[i] = recipe.width(t)
        d.n_src = len(self.series); d.n_runs = len(runs); d.runs = self.runs.data_ptr()
        d.fast_layout = int(fast)      # every row's runs are `history` long (or the single constant-1 run) and follow each other from feature 0
                                       # (fast=False: the general run-by-run gather kernel, kept for descriptors that do not promise this)
        d.n_rows = len(rows); d.rows = self.rows.data_ptr()
        d.n_label = len(recipe.label_cols); d.label_src = sidx[recipe.label_series] if recipe.label_series else 0
        d.label_rotate = int(recipe.label_rotate); d.quat_src = sidx[recipe.quat_series] if recipe.quat_series else -1
        d.label_cols = self.label_cols.data_ptr()
        self.desc = d
        self._cache = {}
        self._src = (C.c_void_p * len(self.series))(*[a.data_ptr() for a in self.series])
        self._pitch = (C.c_int64 * len(self.series))(*[a.shape[1] for a in self.series])     # column stride (rows + slack)
        self._rows = (C.c_int64 * len(self.series))(*[a.shape[1] - 8 for a in self.series])
        self._src16 = None; self._run_ptrs = None

    def __len__(self) -> int:
        """Number of windows (the reference's dataset length: rows - history + 1)."""
        return self.n_rows - self.recipe.history + 1

    def padded_width(self, t: str) -> int:
        return eng.row_pitch(self.recipe.width(t), 2 if self.dtype == "bf16" else 4)      # the engine's input layout

    def _buffers(self, B: int):
        """Output buffers for a batch of B windows, made once per batch size: the pad columns are zeroed here and never
        written again (the kernel only writes feature columns)."""
        if B not in self._cache:
            r = self.recipe
            xs = [torch.zeros(B * r.num_nodes[t], self.padded_width(t), dtype=self.torch_dtype, device=self.device) for t in r.node_types]
            y = torch.empty(B, len(r.label_cols), dtype=torch.float32, device=self.device) if r.label_cols else None
            q = torch.empty(B, 4, dtype=torch.float32, device=self.device) if r.quat_series else None
            self._cache = {B: (xs, y, q)}          # one batch size at a time
        return self._cache[B]

    def assemble(self, starts, reuse_buffers: bool = False) -> Tuple[List[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
        """starts: window start rows (== the reference's dataset indices), a host sequence / tensor (checked on the host) or
        a device int64 tensor (trusted).  Returns (xs, y, r_o): xs[t] is [B * n_t, padded width] at the store's dtype --
        exactly what `Engine.forward` takes (pad columns are zero) --, y float32 [B, n_label], r_o float32 [B, 4].
        reuse_buffers=True returns the same tensors on every call of one batch size (a training loop that consumes the
        batch before asking for the next)."""
        r = self.recipe
        st = starts if isinstance(starts, torch.Tensor) else torch.as_tensor(np.asarray(starts), dtype=torch.int64)
        st = st.flatten().to(torch.int64)
        if st.numel() < 1:
            raise ValueError("no window indices")
        if not st.is_cuda:
            if int(st.min()) < 0 or int(st.max()) + r.history > self.n_rows:
                raise IndexError("window index out of range")
            st = st.to(self.device, non_blocking=True)
        B = st.numel()
        if reuse_buffers:
            xs, y, q = self._buffers(B)
        else:
            xs = [torch.zeros(B * r.num_nodes[t], self.padded_width(t), dtype=self.torch_dtype, device=self.device) for t in r.node_types]
            y = torch.empty(B, len(r.label_cols), dtype=torch.float32, device=self.device) if r.label_cols else None
            q = torch.empty(B, 4, dtype=torch.float32, device=self.device) if r.quat_series else None
        xp = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        pitch = (C.c_int64 * len(xs))(*[x.shape[1] for x in xs])
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self.lib.mshgnn_assemble_windows(C.byref(self.desc), self._src, self._pitch, self._rows, st.data_ptr(), B, xp, pitch,
                                              y.data_ptr() if y is not None else None, q.data_ptr() if q is not None else None, stream)
        eng._check(self.lib, rc, "mshgnn_assemble_windows")
        return xs, y, q

    def batch(self, starts, edge_index_dict) -> "WindowBatch":
        """A minibatch of window indices shaped like the PyG batch the wrappers take (see WindowBatch)."""
        return WindowBatch(self, starts, edge_index_dict)

    def series_step_args(self, bf16: bool = True):
        """What Engine.step_mse_series hands to mshgnn_step_mse_series: bf16 copies of the series (same strides; bf16=False, the split plan:
        none -- it gathers from the fp32 series themselves) and the run-pointer scratch."""
        if self._run_ptrs is None:
            self._run_ptrs = torch.zeros(max(1, int(self.desc.n_runs)), dtype=torch.int64, device=self.device)
        if bf16 and self._src16 is None:
            self.series16 = [a.to(torch.bfloat16) for a in self.series]
            self._src16 = (C.c_void_p * len(self.series16))(*[a.data_ptr() for a in self.series16])
        return (self._src16 if bf16 else None), self._run_ptrs


class WindowBatch:
    """One minibatch of window indices of a `SequenceStore`, with the attributes the reference's wrappers read off a PyG batch
    (`x_dict`, `edge_index_dict`, `y`, `r_o`, `batch_size`; gnnLightning.py:680-722).  Nothing is gathered when it is made: a wrapper's
    `training_step` hands the indices to the engine, whose encoder gathers its inputs from the resident series
    (`models.fused_training_step_windows` -> `mshgnn_step_mse_series` / `mshgnn_step_ce_series`) and leaves the labels here; any other consumer
    (validation, the two-call route) gets the windows assembled on first access (`SequenceStore.assemble`, the store's reusable buffers:
    consume a batch before asking the store for the next one)."""

    def __init__(self, store: SequenceStore, starts, edge_index_dict):
        st = starts if isinstance(starts, torch.Tensor) else torch.as_tensor(np.asarray(starts), dtype=torch.int64)
        st = st.flatten().to(torch.int64)
        # Contract: every start index i satisfies 0 <= i and i + history <= store.n_rows (the fused-gather kernels read the series at
        # [i, i + history) without a bound check).  Host indices are checked here; DEVICE indices are taken as they are, because checking them costs a
        # host synchronisation per batch -- `store.check_starts = True` turns that check on (debugging a sampler).
        if not st.is_cuda or getattr(store, "check_starts", False):
            if st.numel() < 1 or int(st.min()) < 0 or int(st.max()) + store.recipe.history > store.n_rows:
                raise IndexError("window index out of range")
            st = st.to(store.device)
        self.store, self.starts, self.edge_index_dict = store, st, edge_index_dict
        self.batch_size = int(st.numel())
        self._x = self._y = self._q = None

    def _assemble(self):
        if self._x is None:
            xs, y, q = self.store.assemble(self.starts, reuse_buffers=True)
            self._x = dict(zip(self.store.recipe.node_types, xs))
            self._y = y if self._y is None else self._y
            self._q = q if self._q is None else self._q

    def _labels_from_step(self, xs, y, q):
        """(the fused training step materialised the windows and the labels as a by-product)"""
        self._x = dict(zip(self.store.recipe.node_types, xs)) if xs is not None else None
        self._y, self._q = y, q

    @property
    def x_dict(self):
        self._assemble()
        return self._x

    @property
    def y(self):
        if self._y is None:
            self._assemble()
        return self._y

    @property
    def r_o(self):
        if self._q is None:
            self._assemble()
        return self._q
