"""GPU parity tests of the generic-width engine (mshgnn_gen.hip): any hidden multiple of 128, any node count / in-degree -- BASELINE
configs[4] (synthetic 32-limb robot, h=512, 6 layers) and the reference's --hidden_size flag (train_regression-grf_msgn.py:95).
Golden vectors come from the reference's own model files (oracle/gen_golden.py); tolerance 1e-4 relative as for every parity plan."""
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
RTOL = 1e-4
WIDE_CASES = ["synth32_mi_h512_L6_B2", "synth8_mi_h256_L3_B3", "a1c2_h256_L2_d3_B3"]


@pytest.mark.parametrize("dtype", ["f32", "x3"])
@pytest.mark.parametrize("name", WIDE_CASES)
def test_wide_models_match_oracle_and_golden(name, dtype):
    """hidden != 128 / many nodes: plan creation picks the generic engine by itself; a parity request (f32 or x3) runs its split-bf16
    arithmetic."""
    from morphsym_hgnn_amd import engine as eng
    assert torch.cuda.is_available()
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    assert eng.compile_plan_host(spec, dtype).kernel_sets == 4
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype=dtype)
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, f"{name}: stages above {RTOL}: {bad}"
    helpers.check_against_fixture(fx, out, loss, grads, rtol=RTOL, what=name)


@pytest.mark.parametrize("name", [c for c in helpers.GOLDEN_CASES if "_h128_" in c])
def test_every_h128_golden_case_through_the_generic_engine(name, monkeypatch):
    """MSHGNN_ENGINE=generic forces the generic engine for topologies the LDS-resident kernels also cover: every model family (C2, K4,
    MI, the Solo COM variants; regression and classification; base_transform, residual, mean aggregation) at 1e-4."""
    monkeypatch.setenv("MSHGNN_ENGINE", "generic")
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, f"{name}: stages above {RTOL}: {bad}"
    helpers.check_against_fixture(fx, out, loss if spec.regression else None, grads, rtol=RTOL, what=name)


def test_mean_aggregation_with_in_degree_above_one():
    """'mean' relations whose destination nodes have several in-edges (rejected by the LDS-resident kernels; every reference topology has
    degree 1 there): the generic engine scales the aggregated row by 1 / in-degree, as PyG's GraphConv(aggr='mean') does."""
    from morphsym_hgnn_amd import engine as eng, synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    topo = topology.mini_cheetah_k4()
    rel = dict(topo.relations)
    rel[("base", "gt", "base")] = [[0, 1], [2, 1], [3, 1], [1, 0], [2, 0], [0, 3], [1, 2]]      # in-degrees 2, 3, 1, 1
    topo.relations = [(k, rel[k]) for k, _ in topo.relations]
    group, _ = helpers.load_group("mini_cheetah-k4")
    spec = ModelSpec(kind="k4", topology=topo, hidden=128, num_layers=3, widths=synth.feature_widths("k4", True), regression=True, group=group)
    B = 5
    x_dict, y = synth.make_windows(8, B, spec.num_nodes, spec.widths, 4)
    params = synth.make_params(8, spec.param_shapes())
    assert eng.compile_plan_host(spec, "x3").kernel_sets == 4      # the specialised plan rejects it, the generic one takes it
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, topo.edge_index_dict(B), B, dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, bad


def test_aggregate_launch_on_the_32_limb_golden_case(monkeypatch):
    """The aggregate launches forced onto BASELINE configs[4]'s topology (threshold 4 instead of 32 rows, MSHGNN_GEN_MANY, read when the plan is compiled):
    same golden vectors, same tolerance."""
    monkeypatch.setenv("MSHGNN_GEN_MANY", "4")
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("synth32_mi_h512_L6_B2")
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, bad
    helpers.check_against_fixture(fx, out, loss, grads, rtol=RTOL, what="synth32 with aggregate launches")


@pytest.mark.parametrize("dtype", ["x3", "bf16"])
def test_base_node_of_many_limbs_goes_through_the_aggregate_launch(dtype):
    """More than 32 rows into one destination (the base node of a 40-limb robot: 40 hip joints, forward; their 40 gradients, backward): the sums are
    computed by their own launch (k_gagg) per layer and direction, and the forward ones feed the weight-gradient item of that relation as well."""
    from morphsym_hgnn_amd import engine as eng, synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    topo = topology.synthetic_limbs(40)
    # (5 layers: the base node is four hops from the feet, so it is live -- and its 40-row sums exist -- in the first layer only from that depth on)
    spec = ModelSpec(kind="mi", topology=topo, hidden=128, num_layers=5, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=1)
    B = 19
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(21, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(21, spec.param_shapes())
    assert eng.compile_plan_host(spec, dtype).kernel_sets == 4
    tol = RTOL if dtype == "x3" else 3e-2
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, topo.edge_index_dict(B), B, dtype=dtype, **({} if dtype == "x3" else {"decision_tol": 3e-2}))
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad
    e = eng.Engine(spec, dtype)
    xs = e.cast_inputs(x_dict); flat = eng.flatten_params(spec, params, e.device)
    e.profile(True)
    e.step_mse(xs, flat, y.reshape(-1).to(e.device, torch.float32), B)
    torch.cuda.synchronize()
    assert any(r["name"] == "aggregate" and r["launches"] > 0 for r in e.profile_read()), "the aggregate launch ran"


@pytest.mark.parametrize("name", ["synth8_mi_h256_L3_B3", "a1c2_h256_L2_d3_B3", "synth32_mi_h512_L6_B2"])
def test_generic_bf16_arithmetic_is_within_bf16_distance_of_the_oracle(name):
    """The throughput arithmetic of the generic engine (bf16 storage / operands, fp32 accumulate) against the oracle evaluated with the
    engine's relu decisions: every stage within 3e-2 (max-abs relative; bf16 has 8 mantissa bits and the models are 2-6 layers deep), every
    decision that differs from the exact one within 3e-2 of zero; inference == training forward.  The last case is BASELINE configs[4]'s
    model (32 limbs, h = 512, 6 layers: hgnn.py:5-63 with --hidden_size 512, train_regression-grf_msgn.py:95)."""
    from morphsym_hgnn_amd import engine as eng
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    errs, out, *_ = helpers.run_engine_case(spec, x_dict, y, params, ei, B, dtype="bf16", decision_tol=3e-2)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad
    e = eng.Engine(spec, "bf16")
    assert e.generic
    out_inf = e.forward(e.cast_inputs(x_dict), eng.flatten_params(spec, params, e.device), B, training=False)
    assert torch.equal(out_inf.cpu(), out)


@pytest.mark.parametrize("B", [300, 1024])      # 1024: the batch bench.py times configs[4] at
def test_configs4_bench_arithmetic_is_within_bf16_distance_of_the_oracle(B):
    """BASELINE configs[4] at a batch that takes the kernels the bench line is quoted on (>= 256 windows: k_gstep4 on 128-window tiles, the lean
    weight-gradient streams, one ragged tile): the generic engine's bf16 arithmetic at h = 512, L = 6, 129 nodes per window against the fp64 oracle
    evaluated with the engine's relu decisions -- every hidden state, the output, the loss and every parameter gradient within 3e-2 (max-abs
    relative), every differing decision within 3e-2 of zero."""
    from morphsym_hgnn_amd import synth
    case, spec, *_ = helpers.load_case("synth32_mi_h512_L6_B2")
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(41, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(41, spec.param_shapes())
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="bf16", decision_tol=3e-2)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("hidden,kind", [(256, "c2"), (768, "mi"), (1024, "mi")])
def test_pipelined_job_kernel_at_other_widths_is_within_bf16_distance_of_the_oracle(hidden, kind):
    """The widths beside 512 that the pipelined job kernel serves, at a batch that takes it (300 windows: two full 128-window tiles and a ragged one): hidden 256
    (A1-C2 with its residual and base_transform: one 256-column group, the 4-wave form only), 768 (three column groups) and 1024 (two 512-column groups / four of
    256) -- bf16 arithmetic against the fp64 oracle evaluated with the engine's relu decisions, 3e-2 as on the configs[4] case."""
    from morphsym_hgnn_amd import synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    if kind == "c2":
        spec = helpers.make_spec("c2", "a1-c2", "a1-c2", hidden, 2)
    else:
        spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(3), hidden=hidden, num_layers=2, widths={"base": 24, "joint": 9, "foot": 5}, regression=True,
                         grf_dimension=1, group=None, num_timesteps=3)
    B = 300
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(53, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(53, spec.param_shapes())
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="bf16", decision_tol=3e-2)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("seed,nb,nj,nf,hidden,layers", [(1, 1, 7, 3, 128, 3), (2, 2, 15, 5, 128, 2), (3, 3, 30, 9, 256, 3), (4, 1, 5, 2, 128, 4),
                                                        (5, 2, 9, 3, 512, 2), (7, 1, 4, 2, 1024, 2)])      # (seed 6 at 1024: a one-element decoder-bias gradient that cancels to 2e-4 of its terms -- 1.1e-4 on it, 4e-5 elsewhere)
def test_random_topologies_match_the_oracle(seed, nb, nj, nf, hidden, layers):
    """Topologies no robot of the reference has: random edge lists over the five relation types of the MI graph (hgnn.py:5-63 takes any metadata) --
    repeated edges, nodes without in-edges, in-degrees up to the node count, one relation left empty -- on whichever engine plan creation picks
    (the LDS-resident kernels up to 20 nodes at h = 128, else the generic-width engine), parity arithmetic, 1e-4.  hidden = 512 / 1024: the pipelined job kernel
    (k_gstep5, one / two column groups per tile) with a ragged second 64-window tile, sums of two rows as two terms, longer ones through the aggregate launch."""
    import random
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.spec import ModelSpec
    from morphsym_hgnn_amd.topology import RobotTopology
    rng = random.Random(seed)
    n = {"base": nb, "joint": nj, "foot": nf}
    rels = []
    for k, (s_, d_) in enumerate((("base", "joint"), ("joint", "base"), ("joint", "joint"), ("foot", "joint"), ("joint", "foot"))):
        ne = 0 if (k == seed % 5 and k != 4) else rng.randint(1, 2 * max(n[s_], n[d_]))      # (never the relation into the decoder's type: keep a signal path)
        pairs = [[rng.randrange(n[s_]), rng.randrange(n[d_])] for _ in range(ne)]
        if pairs and rng.random() < 0.7:
            pairs.append(list(pairs[0]))                      # a repeated edge: PyG sums it twice
        rels.append(((s_, "connect", d_), pairs))
    topo = RobotTopology(name=f"random{seed}", num_nodes=n, relations=rels)
    spec = ModelSpec(kind="mi", topology=topo, hidden=hidden, num_layers=layers, widths={"base": 24, "joint": 9, "foot": 5}, regression=True,
                     grf_dimension=1, group=None, num_timesteps=3)
    B = 21 if hidden < 512 else 70
    x_dict, y = synth.make_windows(seed, B, spec.num_nodes, spec.widths, spec.out_channels * nf)
    params = synth.make_params(seed, spec.param_shapes())
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, topo.edge_index_dict(B), B, dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, (seed, "generic" if eng.compile_plan_host(spec, "x3").kernel_sets & 4 else "lds-resident", bad)


def test_generic_ragged_batches_and_step_equals_two_call_sequence():
    from morphsym_hgnn_amd import engine as eng, synth
    case, spec, *_ = helpers.load_case("synth8_mi_h256_L3_B3")
    params = synth.make_params(5, spec.param_shapes())
    for B in (1, 17, 65, 130):
        x_dict, y = synth.make_windows(300 + B, B, spec.num_nodes, spec.widths, spec.num_nodes["foot"])
        errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="x3")
        bad = {k: v for k, v in errs.items() if v > RTOL}
        assert not bad, (B, bad)
    e = eng.Engine(spec, "x3")
    B = 130
    x_dict, y = synth.make_windows(9, B, spec.num_nodes, spec.widths, spec.num_nodes["foot"])
    xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32)
    flat = eng.flatten_params(spec, params, e.device)
    out_a = e.forward(xs, flat, B).clone()
    loss_a, g_a = e.backward_mse(xs, flat, out_a, yd, B)
    loss_a, g_a = loss_a.clone(), g_a.clone()
    out_b, loss_b, g_b = e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(g_a, g_b) and torch.equal(loss_a, loss_b)      # deterministic: fixed-order slab sums


def test_job_kernels_of_the_generic_engine_agree_bit_for_bit(monkeypatch):
    """The generic engine's job kernels on the synthetic 32-limb model (h = 512): the 16-wave kernel on 64-window tiles (MSHGNN_GEN_TILE=3, small batches), on
    128-window tiles (k_gstep4, MSHGNN_GEN_TILE=6) and the software-pipelined kernel on the same tiles (k_gstep5 at 8 waves, =8, and at 4 waves with two workgroups
    per CU, =9: the default from 256 windows picks between the two per launch) --
    same operands in the same order per accumulator, so outputs, loss and every gradient are identical bits (full and ragged 128-window tiles)."""
    from morphsym_hgnn_amd import engine as eng, synth
    case, spec, *_ = helpers.load_case("synth32_mi_h512_L6_B2")
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    params = synth.make_params(21, spec.param_shapes())
    for B, modes in ((256, ("3", "6", "8", "9", None)), (300, ("3", "6", "8", "9", None))):
        x_dict, y = synth.make_windows(21 + B, B, spec.num_nodes, spec.widths, n_y)
        res = {}
        for mode in modes:
            if mode is None: monkeypatch.delenv("MSHGNN_GEN_TILE", raising=False)
            else: monkeypatch.setenv("MSHGNN_GEN_TILE", mode)
            e = eng.Engine(spec, "bf16")
            assert e.generic
            xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, params, e.device)
            out, loss, g = e.step_mse(xs, flat, yd, B)
            torch.cuda.synchronize()
            res[mode] = (out.clone(), loss.clone(), g.clone())
        for mode in modes[1:]:
            for i in range(3):
                assert torch.equal(res["3"][i], res[mode][i]), (B, mode, i)
    # split arithmetic: k_gstep at 8 waves (mode 2) and k_gstep4 at 16 waves (the default) against the oracle.  (Not against each other: k_gstep4 stages a single
    # source's hi / lo halves as they are, k_gstep adds them and splits the sum again -- a half-ulp lo half may land on the other side of the hi half's rounding
    # tie, and a 2^-16 perturbation flips relu decisions of pre-activations near zero; each run is compared with the oracle under its own decisions.)
    B = 70
    x_dict, y = synth.make_windows(77, B, spec.num_nodes, spec.widths, n_y)
    for mode in ("2", "6", None):      # k_gstep at 8 waves, k_gstep4, k_gstep5 (the default)
        if mode is None: monkeypatch.delenv("MSHGNN_GEN_TILE", raising=False)
        else: monkeypatch.setenv("MSHGNN_GEN_TILE", mode)
        errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="x3")
        bad = {k: v for k, v in errs.items() if v > RTOL}
        assert not bad, (mode, bad)
