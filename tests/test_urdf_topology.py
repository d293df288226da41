"""URDF -> topology compiler against the reference's own known answers (tests/testGraphParser.py) -- the skeleton
fixtures (tests/golden/urdf_skeletons.json, made by oracle/gen_urdf_skeletons.py) hold only names and parent/child links."""
import json
import os

import numpy as np
import pytest

from morphsym_hgnn_amd import topology, urdf_topology as ut

SKEL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "urdf_skeletons.json")))


def _edge_set(topo, et):
    return sorted(map(tuple, topo.edges(et)))


def test_go1_heterogeneous_known_answers():
    g = ut.HeterogeneousRobotGraph(SKEL["go1"])
    want = {'floating_base': 0, 'FR_hip_joint': 0, 'FR_thigh_joint': 1, 'FR_calf_joint': 2, 'FL_hip_joint': 3, 'FL_thigh_joint': 4,
            'FL_calf_joint': 5, 'RR_hip_joint': 6, 'RR_thigh_joint': 7, 'RR_calf_joint': 8, 'RL_hip_joint': 9, 'RL_thigh_joint': 10,
            'RL_calf_joint': 11, 'FR_foot_fixed': 0, 'FL_foot_fixed': 1, 'RR_foot_fixed': 2, 'RL_foot_fixed': 3}   # testGraphParser.py:294-312
    assert g.get_node_name_to_index_dict() == want
    assert g.get_node_index_to_name_dict('base') == {0: 'floating_base'}                                      # :320-324
    assert g.get_node_index_to_name_dict('foot') == {0: 'FR_foot_fixed', 1: 'FL_foot_fixed', 2: 'RR_foot_fixed', 3: 'RL_foot_fixed'}
    assert g.get_node_index_to_name_dict('joint')[10] == 'RL_thigh_joint'
    assert g.get_num_of_each_node_type() == [1, 12, 4]                                                        # :360
    bj, jb, jj, fj, jf = g.get_edge_index_matrices()                                                          # :370-375
    np.testing.assert_array_equal(bj, [[0, 0, 0, 0], [0, 3, 6, 9]])
    np.testing.assert_array_equal(jb, [[0, 3, 6, 9], [0, 0, 0, 0]])
    np.testing.assert_array_equal(jj, [[0, 1, 1, 2, 3, 4, 4, 5, 6, 7, 7, 8, 9, 10, 10, 11], [1, 0, 2, 1, 4, 3, 5, 4, 7, 6, 8, 7, 10, 9, 11, 10]])
    np.testing.assert_array_equal(fj, [[0, 1, 2, 3], [2, 5, 8, 11]])
    np.testing.assert_array_equal(jf, [[2, 5, 8, 11], [0, 1, 2, 3]])


def test_hyq_normal_graph_known_answers():
    g = ut.NormalRobotGraph(SKEL["hyq"])
    names = ['floating_base', 'lf_haa_joint', 'lf_hfe_joint', 'lf_kfe_joint', 'lf_foot_joint', 'rf_haa_joint', 'rf_hfe_joint',
             'rf_kfe_joint', 'rf_foot_joint', 'lh_haa_joint', 'lh_hfe_joint', 'lh_kfe_joint', 'lh_foot_joint', 'rh_haa_joint',
             'rh_hfe_joint', 'rh_kfe_joint', 'rh_foot_joint']                                                   # testGraphParser.py:28-34
    assert sorted(n.name for n in g.nodes) == sorted(names)
    want_edges = {'trunk_to_lf_haa_joint': ('floating_base', 'lf_haa_joint'), 'trunk_to_rh_haa_joint': ('floating_base', 'rh_haa_joint'),
                  'lf_hipassembly': ('lf_haa_joint', 'lf_hfe_joint'), 'rh_lowerleg': ('rh_kfe_joint', 'rh_foot_joint'),
                  'lh_upperleg': ('lh_hfe_joint', 'lh_kfe_joint')}                                              # :44-66 (sample)
    got = {e.name: (e.parent, e.child) for e in g.edges}
    assert len(got) == 16 and all(got[k] == v for k, v in want_edges.items())
    types = ['base', 'joint', 'joint', 'joint', 'foot'] + ['joint', 'joint', 'joint', 'foot'] * 3             # :84-88
    assert [g.get_node_from_name(n).get_node_type() for n in names] == types
    assert g.get_edge_index_matrix().shape == (2, 32)


def test_xml_round_trip_and_errors():
    xml = ut.skeleton_to_urdf(SKEL["go1"])
    g = ut.HeterogeneousRobotGraph(xml)
    assert g.get_num_of_each_node_type() == [1, 12, 4]
    with pytest.raises(ut.InvalidURDFException):
        ut.RobotGraph({"links": ["a", "b", "lonely"], "joints": [["j", "fixed", "a", "b"]]})


@pytest.mark.parametrize("skel,variant,robot,ref", [
    ("a1_quad_pruned", "c2", "a1", topology.a1_c2), ("mini_cheetah", "c2", "mini_cheetah", topology.mini_cheetah_c2),
    ("mini_cheetah", "k4", "mini_cheetah", topology.mini_cheetah_k4), ("a1_quad_pruned", "mi", "a1", topology.quadruped_mi)])
def test_compiled_topologies_equal_the_hand_written_tables(skel, variant, robot, ref):
    got, want = ut.compile_topology(SKEL[skel], variant, robot), ref()
    assert got.num_nodes == want.num_nodes and got.edge_types == want.edge_types
    for et in want.edge_types:
        assert _edge_set(got, et) == _edge_set(want, et), et


def test_solo_com_graph_shape_the_models_assume():
    """The COM models hard-code 12 joints (hgnn_k4_com.py:29-34); a 12-joint quadruped skeleton compiled without feet
    gives exactly topology.solo_k4_com() (the Solo_ori URDF itself has no floating base -- SURVEY.md section 8(d) config 4)."""
    got = ut.compile_topology(SKEL["mini_cheetah"], "k4", "solo", with_feet=False)
    want = topology.solo_k4_com()
    assert got.num_nodes == want.num_nodes and got.edge_types == want.edge_types
    for et in want.edge_types:
        assert _edge_set(got, et) == _edge_set(want, et), et
    assert ut.HeterogeneousRobotGraph(SKEL["solo12_ori"]).get_num_of_each_node_type() == [4, 8, 4]


def test_synthetic_many_limb_robot():
    g = ut.HeterogeneousRobotGraph(ut.synthetic_limb_robot(32, 3))
    assert g.get_num_of_each_node_type() == [1, 96, 32]
    t = ut.compile_topology(g, "mi", "synthetic")
    assert len(t.edges(("joint", "connect", "joint"))) == 32 * 2 * 2 and len(t.edges(("foot", "connect", "joint"))) == 32
