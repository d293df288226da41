"""URDF -> graph topology compiler (urchin-free): joints become nodes, links become edges.

Reference behaviour restated (relative to /root/reference):
  * src/ms_hgnn/graphParser.py:97-148   links with a parent joint and >= 1 child joint become edges (one per child, named
                                        "<link>_to_<child joint>" when there are several); every joint becomes a node
  * graphParser.py:29-50                node typing: parent edge + child edges -> 'joint', parent only -> 'foot',
                                        children only -> 'base'
  * graphParser.py:305-351,407-446      NormalRobotGraph / HeterogeneousRobotGraph index dictionaries
  * graphParser.py:483-550              the five heterogeneous edge-index matrices bj, jb, jj, fj, jf
  * datasets_py/quadSDKDataset_Morph.py:241-272, LinTzuYaunDataset_Morph.py:410-447,492-523, soloDataset.py:455-512
                                        the per-dataset base-splitting rules of the C2 / K4 graphs
Only names and the parent/child relation of the URDF are read (the reference feeds nothing else of the URDF to the
models: edge attributes are attached but never consumed), so a URDF file, an XML string or
a skeleton dict {"links": [...], "joints": [[name, type, parent_link, child_link], ...]} are all accepted.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .topology import RobotTopology

NODE_TYPES = ["base", "joint", "foot"]


class InvalidURDFException(Exception):
    pass


def load_skeleton(src) -> dict:
    """URDF path / XML string / skeleton dict -> skeleton dict (file order preserved, as urchin does)."""
    if isinstance(src, dict):
        return {"links": list(src["links"]), "joints": [list(j) for j in src["joints"]]}
    text = str(src)
    root = ET.fromstring(text) if text.lstrip().startswith("<") else ET.parse(text).getroot()
    links = [l.attrib["name"] for l in root.findall("link")]
    joints = []
    for j in root.findall("joint"):
        par, chi = j.find("parent"), j.find("child")
        if par is None or chi is None:
            raise InvalidURDFException(f"joint {j.attrib.get('name')} lacks a parent or child link")
        joints.append([j.attrib["name"], j.attrib.get("type", ""), par.attrib["link"], chi.attrib["link"]])
    return {"links": links, "joints": joints}


def skeleton_to_urdf(skel: dict, name: str = "robot") -> str:
    """Minimal URDF text of a skeleton (used by the tests and the synthetic robot generator)."""
    out = [f'<robot name="{name}">']
    out += [f'  <link name="{l}"/>' for l in skel["links"]]
    for jn, jt, par, chi in skel["joints"]:
        out.append(f'  <joint name="{jn}" type="{jt or "revolute"}"><parent link="{par}"/><child link="{chi}"/></joint>')
    out.append("</robot>")
    return "\n".join(out)


@dataclass
class Node:
    name: str
    edge_parent: Optional[str]
    edge_children: List[str]

    def get_node_type(self) -> str:
        if self.edge_parent is not None and self.edge_children:
            return "joint"
        if self.edge_parent is not None:
            return "foot"
        if self.edge_children:
            return "base"
        raise Exception("Every node should have a child or parent edge.")


@dataclass
class Edge:
    name: str
    parent: str
    child: str


class RobotGraph:
    """Joints as nodes, links as edges (graphParser.py:11-148)."""

    def __init__(self, urdf):
        skel = load_skeleton(urdf)
        self.link_names: List[str] = skel["links"]
        self.joints: List[List[str]] = skel["joints"]
        self.edges: List[Edge] = []
        for link in self.link_names:
            parent, children = self._connections_to_link(link)
            if parent is None and not children:
                raise InvalidURDFException("Link connected to no joints.")
            if parent is None or not children:
                continue                                   # cannot be an edge (root / leaf link)
            if len(children) == 1:
                self.edges.append(Edge(link, parent, children[0]))
            else:
                self.edges += [Edge(link + "_to_" + c, parent, c) for c in children]
        self.nodes: List[Node] = []
        for jn, _, jparent, jchild in self.joints:
            ep = None
            for e in self.edges:                           # substring match on the (possibly suffixed) edge name, as the reference
                if jparent in e.name and jn == e.child:
                    ep = e.name
            ec = [e.name for e in self.edges if jchild in e.name and jn == e.parent]
            self.nodes.append(Node(jn, ep, ec))

    def _connections_to_link(self, link: str):
        parent, children = None, []
        for jn, _, jparent, jchild in self.joints:
            if jparent == link:
                children.append(jn)
            elif jchild == link:
                if parent is not None:
                    raise InvalidURDFException("Link has more than two parent joints.")
                parent = jn
        return parent, children

    def get_num_nodes(self) -> int:
        return len(self.nodes)

    def get_node_from_name(self, name: str) -> Optional[Node]:
        for n in self.nodes:
            if n.name == name:
                return n
        return None


class NormalRobotGraph(RobotGraph):
    def get_node_name_to_index_dict(self) -> Dict[str, int]:
        return {n.name: i for i, n in enumerate(self.nodes)}

    def get_node_index_to_name_dict(self) -> Dict[int, str]:
        return {i: n.name for i, n in enumerate(self.nodes)}

    def get_edge_index_matrix(self) -> np.ndarray:
        d = self.get_node_name_to_index_dict()
        cols = []
        for e in self.edges:
            a, b = d[e.parent], d[e.child]
            cols += [[a, b], [b, a]]
        return np.array(cols, dtype=np.int64).T

    def get_edge_connections_to_name_dict(self):
        d = self.get_node_name_to_index_dict()
        out = {}
        for e in self.edges:
            out[(d[e.parent], d[e.child])] = e.name
            out[(d[e.child], d[e.parent])] = e.name
        return out

    def get_edge_name_to_connections_dict(self):
        d = self.get_node_name_to_index_dict()
        return {e.name: np.array([[d[e.parent], d[e.child]], [d[e.child], d[e.parent]]]) for e in self.edges}


class HeterogeneousRobotGraph(RobotGraph):
    def _by_type(self) -> List[List[Node]]:
        return [[n for n in self.nodes if n.get_node_type() == t] for t in NODE_TYPES]

    def get_node_name_to_index_dict(self) -> Dict[str, int]:
        out = {}
        for group in self._by_type():
            out.update({n.name: i for i, n in enumerate(group)})
        return out

    def get_node_name_to_index_dict_for_type(self, type: str) -> Dict[str, int]:
        if type not in NODE_TYPES:
            raise ValueError(type, " is not a valid node type.")
        return {n.name: i for i, n in enumerate(n for n in self.nodes if n.get_node_type() == type)}

    def get_node_index_to_name_dict(self, joint_type: str) -> Dict[int, str]:
        for group in self._by_type():
            if group and group[0].get_node_type() == joint_type:
                return {i: n.name for i, n in enumerate(group)}
        return {}

    def get_num_of_each_node_type(self) -> List[int]:
        return [len(g) for g in self._by_type()]

    def get_edge_index_matrices(self):
        """bj, jb, jj, fj, jf as 2xE integer arrays (graphParser.py:483-550)."""
        d = self.get_node_name_to_index_dict()
        bj, jj, fj = [], [], []
        for e in self.edges:
            pt, ct = self.get_node_from_name(e.parent).get_node_type(), self.get_node_from_name(e.child).get_node_type()
            p, c = d[e.parent], d[e.child]
            if pt == "joint" and ct == "joint":
                jj += [[p, c], [c, p]]
            elif pt == "base" and ct == "joint":
                bj.append([p, c])
            elif pt == "joint" and ct == "foot":
                fj.append([c, p])
            else:
                raise Exception("Not possible")
        m = lambda rows: np.array(rows, dtype=np.int64).reshape(-1, 2).T
        bjm, jjm, fjm = m(bj), m(jj), m(fj)
        return bjm, bjm[[1, 0]], jjm, fjm, fjm[[1, 0]]


# ------------------------------------------------------------------------------------------------------------------
# URDF graph -> RobotTopology of the model variants
# ------------------------------------------------------------------------------------------------------------------
# base-splitting rules of the symmetric graphs: (front hips, back hips) per base node, as joint indices
_C2_RULES = {
    "a1": ([0, 6], [3, 9]),              # quadSDKDataset_Morph.py:253-258 (joint order FL, RL, FR, RR)
    "mini_cheetah": ([3, 9], [0, 6]),    # LinTzuYaunDataset_Morph.py:506-509 (also soloDataset.py:496-499)
    "solo": ([3, 9], [0, 6]),
}
_K4_RULES = {
    "mini_cheetah": ([[0, 1], [1, 0], [2, 3], [3, 2]], [[0, 2], [2, 0], [1, 3], [3, 1]]),   # gt, gs: LinTzuYaunDataset_Morph.py:432-435
    "solo": ([[0, 2], [2, 0], [1, 3], [3, 1]], [[0, 1], [1, 0], [2, 3], [3, 2]]),           # soloDataset.py:476-479 (names swapped)
}


def _pairs(m: np.ndarray) -> List[List[int]]:
    return [[int(a), int(b)] for a, b in m.T]


def compile_topology(urdf, variant: str = "mi", robot: str = "a1", with_feet: bool = True, name: Optional[str] = None) -> RobotTopology:
    """Topology of one window for `variant` in {'mi', 'c2', 'k4'}: 'mi' is the parser's own heterogeneous graph
    (flexibleDataset.py:308-322); 'c2' / 'k4' replace the single base by 2 / 4 base nodes with the dataset's rules.
    with_feet=False drops the foot type (Solo centroidal-momentum graphs, soloDataset.py:201-233)."""
    g = urdf if isinstance(urdf, HeterogeneousRobotGraph) else HeterogeneousRobotGraph(urdf)
    nb, nj, nf = g.get_num_of_each_node_type()
    bj, jb, jj, fj, jf = g.get_edge_index_matrices()
    rel: List[Tuple[Tuple[str, str, str], List[List[int]]]] = []
    if variant == "mi":
        nodes = {"base": nb, "joint": nj}
        rel += [(("base", "connect", "joint"), _pairs(bj)), (("joint", "connect", "base"), _pairs(jb)),
                (("joint", "connect", "joint"), _pairs(jj))]
    elif variant == "c2":
        if robot not in _C2_RULES:
            raise ValueError(f"no C2 base-splitting rule for robot {robot!r}")
        front, back = _C2_RULES[robot]
        nodes = {"base": 2, "joint": nj}
        rel += [(("base", "front_bj", "joint"), [[b, j] for b, j in enumerate(front)]),
                (("joint", "front_bj", "base"), [[j, b] for b, j in enumerate(front)]),
                (("base", "back_bj", "joint"), [[b, j] for b, j in enumerate(back)]),
                (("joint", "back_bj", "base"), [[j, b] for b, j in enumerate(back)]),
                (("joint", "connect", "joint"), _pairs(jj))]
    elif variant == "k4":
        if robot not in _K4_RULES:
            raise ValueError(f"no K4 base-splitting rule for robot {robot!r}")
        gt, gs = _K4_RULES[robot]
        nodes = {"base": 4, "joint": nj}
        rel += [(("base", "connect", "joint"), [[b, 3 * b] for b in range(4)]),
                (("joint", "connect", "base"), [[3 * b, b] for b in range(4)]),
                (("joint", "connect", "joint"), _pairs(jj))]
    else:
        raise ValueError(f"unknown variant {variant!r}")
    if with_feet:
        nodes["foot"] = nf
        rel += [(("foot", "connect", "joint"), _pairs(fj)), (("joint", "connect", "foot"), _pairs(jf))]
    if variant == "c2":
        rel.append((("base", "center_bb", "base"), [[0, 1], [1, 0]]))
    if variant == "k4":
        rel += [(("base", "gt", "base"), gt), (("base", "gs", "base"), gs)]
    return RobotTopology(name=name or f"{robot}-{variant}", num_nodes=nodes, relations=rel)


def synthetic_limb_robot(num_limbs: int = 32, joints_per_limb: int = 3) -> dict:
    """Skeleton of a floating-base robot with `num_limbs` identical serial limbs ending in a fixed foot joint (the
    synthetic many-limb robot of SURVEY.md section 8(d) config 5)."""
    links, joints = ["world", "trunk"], [["floating_base", "floating", "world", "trunk"]]
    for l in range(num_limbs):
        prev = "trunk"
        for k in range(joints_per_limb):
            link = f"limb{l}_link{k}"
            links.append(link)
            joints.append([f"limb{l}_joint{k}", "revolute", prev, link])
            prev = link
        links.append(f"limb{l}_foot")
        joints.append([f"limb{l}_foot_fixed", "fixed", prev, f"limb{l}_foot"])
    return {"links": links, "joints": joints}
