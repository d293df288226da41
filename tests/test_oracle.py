"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors (generated from the reference
import, oracle/gen_golden.py), the exact group-equivariance identity, topology tables and parameter counts."""
import numpy as np
import pytest
import torch

from morphsym_hgnn_amd import synth, topology
from oracle import ms_hgnn_oracle as orc
from tests import helpers


@pytest.mark.parametrize("name", helpers.GOLDEN_CASES)
def test_oracle_reproduces_golden(name):
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    assert spec.num_params() == int(fx["n_params"])
    out, loss, grads = orc.step(helpers.oracle_config(spec), params, x_dict, ei, y, case["B"])
    helpers.check_against_fixture(fx, out, loss, grads, rtol=1e-11, what=name)


def test_parameter_counts_match_reference():
    # 996 227 (C2 h128 L3 d3), 2 312 067 (C2 L8), 2 144 642 (K4 cls L8) -- probe-verified
    assert helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3).num_params() == 996227
    assert helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 8).num_params() == 2312067
    assert helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 8, regression=False).num_params() == 2144642


def test_quadruped_tables_match_graph_parser_vectors():
    # /root/reference/tests/testGraphParser.py:360-375 (Go1): node counts and bj/jb/jj/fj/jf matrices
    t = topology.quadruped_mi()
    assert [t.num_nodes[k] for k in ("base", "joint", "foot")] == [1, 12, 4]
    as_mat = lambda et: np.array(t.edges(et)).T
    np.testing.assert_array_equal(as_mat(("base", "connect", "joint")), [[0, 0, 0, 0], [0, 3, 6, 9]])
    np.testing.assert_array_equal(as_mat(("joint", "connect", "base")), [[0, 3, 6, 9], [0, 0, 0, 0]])
    np.testing.assert_array_equal(as_mat(("joint", "connect", "joint")),
                                  [[0, 1, 1, 2, 3, 4, 4, 5, 6, 7, 7, 8, 9, 10, 10, 11],
                                   [1, 0, 2, 1, 4, 3, 5, 4, 7, 6, 8, 7, 10, 9, 11, 10]])
    np.testing.assert_array_equal(as_mat(("foot", "connect", "joint")), [[0, 1, 2, 3], [2, 5, 8, 11]])
    np.testing.assert_array_equal(as_mat(("joint", "connect", "foot")), [[2, 5, 8, 11], [0, 1, 2, 3]])


def test_batched_edge_index_roundtrip():
    t = topology.a1_c2()
    ei = t.edge_index_dict(5)
    for (s, r, d), pairs in t.relations:
        got = topology.infer_window_edges(ei[(s, r, d)], t.num_nodes[s], t.num_nodes[d], 5)
        assert got == pairs
    bad = ei[("joint", "connect", "joint")].clone()
    bad[0, -1] -= 1
    with pytest.raises(ValueError):
        topology.infer_window_edges(bad, 12, 12, 5)


def _permute_nodes(x, n, perm):
    return x.view(-1, n, x.shape[1])[:, perm, :].reshape(x.shape)


def _act_c2(group, x_dict, T=150):
    """g_s acting on an A1-C2 window: joints permuted by permutation_Q_js[0] with sign
    reflection_Q_js[0] on every variable; the two base nodes swapped with lin/ang axis reflections."""
    pj = torch.tensor(group["permutation_Q_js"][0])
    rj = torch.tensor(group["reflection_Q_js"][0], dtype=torch.float64)
    j = x_dict["joint"].view(-1, 12, x_dict["joint"].shape[1])
    j = (j[:, pj, :] * rj.view(1, 12, 1)).reshape(x_dict["joint"].shape)
    b = x_dict["base"].view(-1, 2, 6, T)
    # permutation_Q_bs[0] swaps the two base nodes' xyz triples; reflections act per axis
    rl = torch.tensor(group["reflection_Q_bs_lin"][0], dtype=torch.float64).view(2, 3)
    ra = torch.tensor(group["reflection_Q_bs_ang"][0], dtype=torch.float64).view(2, 3)
    refl = torch.cat((rl, ra), dim=1).view(1, 2, 6, 1)
    b = (b[:, [1, 0], :, :] * refl).reshape(x_dict["base"].shape)
    return {"base": b, "joint": j, "foot": x_dict["foot"]}


def test_c2_equivariance_identity_is_exact():
    """f(g.x) == g.f(x) with max-abs-diff exactly 0.0 in float64; the negative control with
    symmetry_mode=None must break it -- so masks, weight sharing and topology are all exercised."""
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 4
    x_dict, _ = synth.make_windows(11, B, spec.num_nodes, spec.widths, 12)
    # physically consistent C2 input: both base nodes carry the same IMU window (np.tile, quadSDKDataset_Morph.py:109)
    params = synth.make_params(11, spec.param_shapes())
    ei = spec.topology.edge_index_dict(B)
    cfg = helpers.oracle_config(spec)
    out = orc.forward(cfg, params, x_dict, ei)
    out_g = orc.forward(cfg, params, _act_c2(spec.group, x_dict), ei)
    pf = torch.tensor(spec.group["permutation_Q_fs"][0])
    rf = torch.tensor(spec.group["reflection_Q_fs"][0], dtype=torch.float64)
    g_out = out[:, pf] * rf
    assert float((out_g - g_out).abs().max()) == 0.0
    # negative control
    spec0 = helpers.make_spec("c2", "a1-c2", None, 128, 3)
    cfg0 = helpers.oracle_config(spec0)
    o0 = orc.forward(cfg0, params, x_dict, ei)
    o0g = orc.forward(cfg0, params, _act_c2(spec.group, x_dict), ei)
    assert float((o0g - o0[:, pf] * rf).abs().max()) > 1e-3


def test_masks_match_between_spec_and_oracle():
    for kind, topo, cfg, reg in [("c2", "a1-c2", "a1-c2", True), ("c2", "mini_cheetah-c2", "mini_cheetah-c2", False),
                                 ("k4", "mini_cheetah-k4", "mini_cheetah-k4", False)]:
        spec = helpers.make_spec(kind, topo, cfg, 128, 1, regression=reg)
        om = orc.input_masks(helpers.oracle_config(spec), spec.num_nodes, spec.widths)
        sm = spec.input_masks()
        for t, m in om.items():
            assert torch.equal(m, sm[t])
        for t in sm:
            if t not in om:
                assert bool((sm[t] == 1).all())
