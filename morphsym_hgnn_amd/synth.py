"""Deterministic synthetic windows and weights for the MS-HGNN path (no dataset, no network).

A counter-based generator (splitmix64) makes every tensor a pure function of (seed, tag, shape), so the
golden-vector generator (build container) and the parity tests / bench (GPU box) produce bit-identical
float64 inputs and weights without shipping megabytes of fixtures.

Shapes follow SURVEY.md section 8(d): A1-C2 windows have base [B,2,900] (one IMU window, axis-major
`flatten('F')`, tiled to both base nodes -- quadSDKDataset_Morph.py:109-110,333), joint [B,12,450],
foot ones [B,4,1] (flexibleDataset.py:563-565), labels [B,12].
"""
from __future__ import annotations

import zlib
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def det_uniform(seed: int, tag: str, shape: Sequence[int], lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """float64 tensor of `shape`, uniform in [lo, hi), pure function of (seed, tag)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((seed * 0x100000001B3 + zlib.crc32(tag.encode())) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + (base << np.uint64(20))) & _MASK
    u = (_splitmix64(_splitmix64(ctr) ^ base) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return torch.from_numpy(lo + (hi - lo) * u).reshape(tuple(shape))


def det_normal(seed: int, tag: str, shape: Sequence[int], mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """Approximately normal (sum of 4 uniforms, variance-matched); exact distribution is irrelevant,
    determinism is what matters."""
    s = sum(det_uniform(seed, f"{tag}#{k}", shape, -1.0, 1.0) for k in range(4))
    return mean + std * s * (3.0 / 4.0) ** 0.5


def feature_widths(kind: str, regression: bool = True, T: int = 150) -> Dict[str, int]:
    """Per-type input width.  A1-C2 regression: 900/450/1 (quadSDKDataset_Morph.py:444-488);
    MiniCheetah (K4, C2-classification): 900/300/900 (LinTzuYaunDataset.py:79-88);
    MI-HGNN on A1: 900/450/1 with a single base node."""
    if kind in ("k4_com", "c2_com", "s4_com"):       # Solo COM: T=1, joint (q, qd), base 6 (soloDataset.py:382-400)
        return {"base": 6, "joint": 2}
    if kind == "k4" or (kind == "c2" and not regression):
        return {"base": 6 * T, "joint": 2 * T, "foot": 6 * T}
    return {"base": 6 * T, "joint": 3 * T, "foot": 1}


def make_windows(seed: int, batch_size: int, num_nodes: Dict[str, int], widths: Dict[str, int],
                 out_width: int, physical_scale: bool = False, classification: bool = False
                 ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """Synthetic minibatch in the reference's calling convention: x_dict[type] is [B*n_type, F_type]
    (graph-major), y is [B, out_width] float64 (labels in {0,1} when classifying)."""
    B = batch_size
    T6 = widths["base"]
    imu = det_normal(seed, "imu", (B, 1, T6))
    if physical_scale:
        T = T6 // 6
        imu = imu.clone()
        imu[:, :, 2 * T:3 * T] += 9.8          # gravity on lin-acc z
        imu[:, :, 3 * T:] *= 0.5               # ang vel
    base = imu.expand(B, num_nodes["base"], T6).reshape(B * num_nodes["base"], T6).clone()
    joint = det_normal(seed, "joint", (B * num_nodes["joint"], widths["joint"]), std=2.0 if physical_scale else 1.0)
    if "foot" not in num_nodes:
        foot = None
    elif widths["foot"] == 1:
        foot = torch.ones(B * num_nodes["foot"], 1, dtype=torch.float64)
    else:
        foot = det_normal(seed, "foot", (B * num_nodes["foot"], widths["foot"]))
    if classification:
        y = (det_uniform(seed, "y", (B, out_width)) > 0).to(torch.float64)
    else:
        y = det_normal(seed, "y", (B, out_width), std=30.0 if physical_scale else 1.0)
    x = {"base": base, "joint": joint}
    if foot is not None:
        x["foot"] = foot
    return x, y


def make_params(seed: int, shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, torch.Tensor]:
    """Module-default-like init (U(+-1/sqrt(fan_in)) for weights and biases), deterministic per name.
    `shapes` maps state_dict names to shapes; bias fan_in is taken from the sibling weight."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith(".weight"):
            fan_in = shape[1]
        else:
            w = name[: -len("bias")] + "weight"
            fan_in = shapes[w][1]
        bound = 1.0 / float(fan_in) ** 0.5
        out[name] = det_uniform(seed, name, shape, -bound, bound)
    return out
