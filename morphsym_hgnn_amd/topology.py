"""Fixed per-robot graph topologies of the MS-HGNN path.

Every time-window graph in a minibatch shares one tiny topology, so the engine
compiles it once into a plan.  This module states those topologies in the
reference's vocabulary (node types ``base``/``joint``/``foot``, relations as
``(src, rel, dst)`` triples) and builds the PyG-style batched ``edge_index_dict``
the reference's ``forward(x_dict, edge_index_dict)`` receives.

Reference anchors (relative to /root/reference):
  * C2 graph, A1 (Quad-SDK):   src/ms_hgnn/datasets_py/quadSDKDataset_Morph.py:241-272, 274-302
  * C2/K4 graph, MiniCheetah:  src/ms_hgnn/datasets_py/LinTzuYaunDataset_Morph.py:410-447, 492-523, 525-550
  * MI-HGNN baseline graph:    src/ms_hgnn/datasets_py/flexibleDataset.py:308-322, graphParser.py:483-550
  * quadruped bj/jj/fj tables: tests/testGraphParser.py:360-375
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import torch

EdgeType = Tuple[str, str, str]

NODE_TYPES = ["base", "joint", "foot"]


def _leg_chain_edges(num_legs: int = 4, joints_per_leg: int = 3):
    """joint<->joint chain edges of a legged robot, in the order the URDF parser emits them
    (tests/testGraphParser.py:372: 0-1, 1-0, 1-2, 2-1, 3-4, ...)."""
    src, dst = [], []
    for leg in range(num_legs):
        b = leg * joints_per_leg
        for k in range(joints_per_leg - 1):
            src += [b + k, b + k + 1]
            dst += [b + k + 1, b + k]
    # the parser lists, per joint, its neighbours in ascending order:
    pairs = sorted(zip(src, dst))
    return [list(p) for p in pairs]


def _foot_joint_edges(num_legs: int = 4, joints_per_leg: int = 3):
    """foot i hangs off the last joint of leg i (tests/testGraphParser.py:374-375)."""
    return [[i, i * joints_per_leg + joints_per_leg - 1] for i in range(num_legs)]


@dataclass
class RobotTopology:
    """One window's graph: node counts per type and an ordered relation list."""

    name: str
    num_nodes: Dict[str, int]
    # ordered like the dataset's get_data_metadata(); each value is a list of [src, dst] pairs
    relations: List[Tuple[EdgeType, List[List[int]]]] = field(default_factory=list)

    @property
    def node_types(self) -> List[str]:
        return [t for t in NODE_TYPES if t in self.num_nodes]

    @property
    def edge_types(self) -> List[EdgeType]:
        return [et for et, _ in self.relations]

    def metadata(self):
        """Same shape as the dataset's ``get_data_metadata()`` (node_types, edge_types)."""
        return self.node_types, self.edge_types

    def edges(self, et: EdgeType) -> List[List[int]]:
        for k, e in self.relations:
            if k == tuple(et):
                return e
        raise KeyError(et)

    def edge_index_dict(self, batch_size: int, device=None) -> Dict[EdgeType, torch.Tensor]:
        """PyG collate semantics: per-graph indices offset by g*n_src / g*n_dst, graph-major."""
        out = {}
        for (s, r, d), pairs in self.relations:
            e = torch.tensor(pairs, dtype=torch.long).t().contiguous()  # [2, E]
            if e.numel() == 0:
                e = e.reshape(2, 0)
            off = torch.arange(batch_size, dtype=torch.long).view(batch_size, 1, 1)
            scale = torch.tensor([self.num_nodes[s], self.num_nodes[d]], dtype=torch.long).view(1, 2, 1)
            b = (e.unsqueeze(0) + off * scale).permute(1, 0, 2).reshape(2, -1)
            out[(s, r, d)] = b.to(device) if device is not None else b
        return out


def a1_c2() -> RobotTopology:
    """A1 (Quad-SDK sim) C2 graph: 2 base + 12 joint + 4 foot, 8 relations.
    quadSDKDataset_Morph.py:253-262 (edges), :291-300 (relation order).  Joint order FL,RL,FR,RR."""
    jj = _leg_chain_edges()
    fj = _foot_joint_edges()
    return RobotTopology(
        name="a1-c2",
        num_nodes={"base": 2, "joint": 12, "foot": 4},
        relations=[
            (("base", "front_bj", "joint"), [[0, 0], [1, 6]]),
            (("joint", "front_bj", "base"), [[0, 0], [6, 1]]),
            (("base", "back_bj", "joint"), [[0, 3], [1, 9]]),
            (("joint", "back_bj", "base"), [[3, 0], [9, 1]]),
            (("joint", "connect", "joint"), jj),
            (("foot", "connect", "joint"), fj),
            (("joint", "connect", "foot"), [[j, f] for f, j in fj]),
            (("base", "center_bb", "base"), [[0, 1], [1, 0]]),
        ],
    )


def mini_cheetah_c2() -> RobotTopology:
    """MiniCheetah C2 graph (LinTzuYaunDataset_Morph.py:506-512): front/back hips swapped vs A1."""
    t = a1_c2()
    t.name = "mini_cheetah-c2"
    rel = dict(t.relations)
    rel[("base", "front_bj", "joint")] = [[0, 3], [1, 9]]
    rel[("joint", "front_bj", "base")] = [[3, 0], [9, 1]]
    rel[("base", "back_bj", "joint")] = [[0, 0], [1, 6]]
    rel[("joint", "back_bj", "base")] = [[0, 0], [6, 1]]
    t.relations = [(k, rel[k]) for k, _ in t.relations]
    return t


def mini_cheetah_k4() -> RobotTopology:
    """MiniCheetah K4 graph: 4 base + 12 joint + 4 foot, 7 relations.
    LinTzuYaunDataset_Morph.py:425-435 (edges), :532-540 (relation order)."""
    jj = _leg_chain_edges()
    fj = _foot_joint_edges()
    bj = [[b, 3 * b] for b in range(4)]
    return RobotTopology(
        name="mini_cheetah-k4",
        num_nodes={"base": 4, "joint": 12, "foot": 4},
        relations=[
            (("base", "connect", "joint"), bj),
            (("joint", "connect", "base"), [[j, b] for b, j in bj]),
            (("joint", "connect", "joint"), jj),
            (("foot", "connect", "joint"), fj),
            (("joint", "connect", "foot"), [[j, f] for f, j in fj]),
            (("base", "gt", "base"), [[0, 1], [1, 0], [2, 3], [3, 2]]),
            (("base", "gs", "base"), [[0, 2], [2, 0], [1, 3], [3, 1]]),
        ],
    )


def quadruped_mi() -> RobotTopology:
    """MI-HGNN baseline graph: 1 base + 12 joint + 4 foot, 5 relations
    (flexibleDataset.py:316-321; tests/testGraphParser.py:370-375)."""
    jj = _leg_chain_edges()
    fj = _foot_joint_edges()
    bj = [[0, 3 * leg] for leg in range(4)]
    return RobotTopology(
        name="quadruped-mi",
        num_nodes={"base": 1, "joint": 12, "foot": 4},
        relations=[
            (("base", "connect", "joint"), bj),
            (("joint", "connect", "base"), [[j, b] for b, j in bj]),
            (("joint", "connect", "joint"), jj),
            (("foot", "connect", "joint"), fj),
            (("joint", "connect", "foot"), [[j, f] for f, j in fj]),
        ],
    )


def solo_k4_com() -> RobotTopology:
    """Solo-12 K4 graph of the centroidal-momentum task: 4 base + 12 joint, no foot type, 5 relations
    (soloDataset.py:207-214 relation order, :455-487 edges; note gs/gt pairs are swapped vs MiniCheetah)."""
    bj = [[b, 3 * b] for b in range(4)]
    return RobotTopology(
        name="solo-k4-com",
        num_nodes={"base": 4, "joint": 12},
        relations=[
            (("base", "connect", "joint"), bj),
            (("joint", "connect", "base"), [[j, b] for b, j in bj]),
            (("joint", "connect", "joint"), _leg_chain_edges()),
            (("base", "gt", "base"), [[0, 2], [2, 0], [1, 3], [3, 1]]),
            (("base", "gs", "base"), [[0, 1], [1, 0], [2, 3], [3, 2]]),
        ],
    )


def solo_c2_com() -> RobotTopology:
    """Solo-12 C2 graph of the COM task: 2 base + 12 joint (soloDataset.py:224-232, :489-512)."""
    return RobotTopology(
        name="solo-c2-com",
        num_nodes={"base": 2, "joint": 12},
        relations=[
            (("base", "front_bj", "joint"), [[0, 3], [1, 9]]),
            (("joint", "front_bj", "base"), [[3, 0], [9, 1]]),
            (("base", "back_bj", "joint"), [[0, 0], [1, 6]]),
            (("joint", "back_bj", "base"), [[0, 0], [6, 1]]),
            (("joint", "connect", "joint"), _leg_chain_edges()),
            (("base", "center_bb", "base"), [[0, 1], [1, 0]]),
        ],
    )


def solo_s4_com() -> RobotTopology:
    """Solo-12 baseline graph of the COM task: 1 base + 12 joint, 3 relations (soloDataset.py:215-220)."""
    bj = [[0, 3 * leg] for leg in range(4)]
    return RobotTopology(
        name="solo-s4-com",
        num_nodes={"base": 1, "joint": 12},
        relations=[
            (("base", "connect", "joint"), bj),
            (("joint", "connect", "base"), [[j, b] for b, j in bj]),
            (("joint", "connect", "joint"), _leg_chain_edges()),
        ],
    )


def synthetic_limbs(num_limbs: int = 32, joints_per_limb: int = 3) -> RobotTopology:
    """MI-HGNN graph of the synthetic many-limb robot (BASELINE.json configs[4] / SURVEY.md 8(d) config 5): 1 base,
    num_limbs x joints_per_limb joints, num_limbs feet, the 5 relations of the quadruped graph -- compiled from the
    generated URDF skeleton by the same parser rules as every other robot (urdf_topology.compile_topology; at 4 limbs
    it reproduces quadruped_mi() exactly)."""
    from . import urdf_topology as ut      # (imports this module)
    return ut.compile_topology(ut.synthetic_limb_robot(num_limbs, joints_per_limb), "mi", name=f"synth{num_limbs}-mi")


TOPOLOGIES = {
    "a1-c2": a1_c2,
    "mini_cheetah-c2": mini_cheetah_c2,
    "mini_cheetah-k4": mini_cheetah_k4,
    "quadruped-mi": quadruped_mi,
    "solo-k4-com": solo_k4_com,
    "solo-c2-com": solo_c2_com,
    "solo-s4-com": solo_s4_com,
    "synth32-mi": lambda: synthetic_limbs(32),
    "synth8-mi": lambda: synthetic_limbs(8),
}


def infer_window_edges(edge_index: torch.Tensor, n_src: int, n_dst: int, batch_size: int) -> List[List[int]]:
    """Recover the per-window edge list from a PyG-batched ``edge_index`` and verify that the batch
    really is ``batch_size`` copies of one topology (the contract the dense layout relies on,
    tests/testGnnLightning.py:243-281).  Raises ValueError otherwise."""
    ei = edge_index.detach().to("cpu", torch.long)
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(ei.shape)}")
    E = ei.shape[1]
    if batch_size <= 0 or E % batch_size != 0:
        raise ValueError(f"edge count {E} is not a multiple of batch size {batch_size}")
    e = E // batch_size
    per = ei.view(2, batch_size, e)
    off = torch.arange(batch_size).view(1, batch_size, 1) * torch.tensor([n_src, n_dst]).view(2, 1, 1)
    local = per - off
    if not bool((local == local[:, :1, :]).all()):
        raise ValueError("edge_index is not a batch of identical window graphs")
    first = local[:, 0, :]
    if e and (first.min() < 0 or first[0].max() >= n_src or first[1].max() >= n_dst):
        raise ValueError("edge_index refers to nodes outside its window")
    return [[int(first[0, k]), int(first[1, k])] for k in range(e)]
