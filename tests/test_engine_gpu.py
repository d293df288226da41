"""GPU parity tests (-m gpu): the HIP engine through the C-ABI vs the fp64 oracle and the golden vectors.

Tolerance (BASELINE.json north_star): 1e-4 relative fp32 for the fp32 plan.  "Relative" = max-abs error over a
tensor / max-abs of the reference tensor (outputs, hidden states, each parameter gradient)."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL_F32 = 1e-4


def _require_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to fall through to)")


@pytest.mark.parametrize("name", [c for c in helpers.GOLDEN_CASES if "_h128_" in c])
def test_engine_matches_oracle_and_golden_f32(name):
    _require_gpu()
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="f32")
    bad = {k: v for k, v in errs.items() if v > RTOL_F32}
    assert not bad, f"{name}: stages above {RTOL_F32}: {bad}"
    # and against the committed vectors generated from the reference import
    helpers.check_against_fixture(fx, out, loss if spec.regression else None, grads, rtol=RTOL_F32, what=name)


def test_dead_parameters_get_exact_zero_gradients():
    """Last-layer relations into base/joint cannot influence the foot output: the reference leaves their
    .grad None, the engine writes exact zeros."""
    _require_gpu()
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    _, _, _, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="f32")
    k = "convs.2.convs.<base___front_bj___joint>.lin_rel.weight"
    assert float(grads[k].abs().max()) == 0.0
    assert float(fx["gnorm:" + k]) == 0.0


def test_ragged_batches_and_repeatability():
    """Batch sizes that are not a multiple of the window tile (16), incl. B=1; two runs are bit-identical
    (slab reduction is deterministic, no float atomics on the gradient path)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    from oracle import ms_hgnn_oracle as orc
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    cfg = helpers.oracle_config(spec)
    e = eng.Engine(spec, "f32")
    params = synth.make_params(5, spec.param_shapes())
    flat = eng.flatten_params(spec, params, e.device)
    for B in (1, 15, 17, 33):
        x_dict, y = synth.make_windows(100 + B, B, spec.num_nodes, spec.widths, 12)
        ei = spec.topology.edge_index_dict(B)
        o_out, o_loss, o_grads = orc.step(cfg, params, x_dict, ei, y, B)
        xs = e.cast_inputs(x_dict)
        out = e.forward(xs, flat, B)
        loss, g = e.mse_loss(out.view(-1), y.reshape(-1).to(e.device, torch.float32))
        gf1 = e.backward(xs, flat, g, B).clone()
        out2 = e.forward(xs, flat, B)
        gf2 = e.backward(xs, flat, g, B)
        torch.cuda.synchronize()
        assert torch.equal(out, out2) and torch.equal(gf1, gf2)
        assert float((out.cpu().double().view(-1) - o_out.reshape(-1)).abs().max() / o_out.abs().max()) < RTOL_F32
        grads = eng.unflatten(spec, gf1.cpu())
        for k, ref in o_grads.items():
            m = float(ref.abs().max())
            if m == 0:
                assert float(grads[k].abs().max()) == 0.0
            else:
                assert float((grads[k].double() - ref).abs().max()) / m < RTOL_F32, (B, k)


def test_equivariance_identity_on_gpu():
    """f(g.x) == g.f(x) through the HIP path.  Exact in fp64; in fp32 the two sides round
    differently only through reassociation, so they agree to 1e-5 relative."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    from tests.test_oracle import _act_c2
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 8
    x_dict, _ = synth.make_windows(11, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(11, spec.param_shapes())
    e = eng.Engine(spec, "f32")
    flat = eng.flatten_params(spec, params, e.device)
    out = e.forward(e.cast_inputs(x_dict), flat, B, training=False).view(B, 12).cpu().double()
    out_g = e.forward(e.cast_inputs(_act_c2(spec.group, x_dict)), flat, B, training=False).view(B, 12).cpu().double()
    pf = torch.tensor(spec.group["permutation_Q_fs"][0])
    rf = torch.tensor(spec.group["reflection_Q_fs"][0], dtype=torch.float64)
    assert float((out_g - out[:, pf] * rf).abs().max() / out.abs().max()) < 1e-5


def test_full_size_batch_properties():
    """BASELINE config size (B=8192): linearity of the gradient in grad_out and batch-additivity -- properties
    that do not need the oracle at full size."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 8192
    e = eng.Engine(spec, "f32")
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(B * spec.num_nodes[t], spec.widths[t], generator=g).to(e.device) for t in spec.node_types]
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    out = e.forward(xs, flat, B)
    go = torch.randn(B * 4, 3, generator=g).to(e.device)
    g1 = e.backward(xs, flat, go, B).clone()
    g2 = e.backward(xs, flat, 2.0 * go, B).clone()
    assert float((g2 - 2.0 * g1).abs().max() / g1.abs().max()) < 1e-5
    # batch additivity: the gradient of the two halves adds up to the gradient of the whole
    h = B // 2
    xa = [x.view(B, -1)[:h].reshape(h * spec.num_nodes[t], -1).contiguous() for x, t in zip(xs, spec.node_types)]
    xb = [x.view(B, -1)[h:].reshape(h * spec.num_nodes[t], -1).contiguous() for x, t in zip(xs, spec.node_types)]
    oa = e.forward(xa, flat, h)
    ga = e.backward(xa, flat, go[: h * 4].contiguous(), h).clone()
    ob = e.forward(xb, flat, h)
    gb = e.backward(xb, flat, go[h * 4:].contiguous(), h).clone()
    assert torch.allclose(torch.cat([oa, ob]), out, rtol=0, atol=0)
    assert float((ga + gb - g1).abs().max() / g1.abs().max()) < 1e-4


def test_fused_mse_backward_equals_two_step():
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 37
    e = eng.Engine(spec, "f32")
    x_dict, y = synth.make_windows(9, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, synth.make_params(9, spec.param_shapes()), e.device)
    yd = y.reshape(-1).to(e.device, torch.float32)
    out = e.forward(xs, flat, B)
    loss1, g = e.mse_loss(out.view(-1), yd)
    g1 = e.backward(xs, flat, g, B).clone()
    out = e.forward(xs, flat, B)
    loss2, g2 = e.backward_mse(xs, flat, out, yd, B)
    torch.cuda.synchronize()
    assert abs(float(loss1) - float(loss2)) / float(loss1) < 1e-5
    assert float((g1 - g2).abs().max() / g1.abs().max()) < 1e-6


@pytest.mark.parametrize("route", ["f32", "x3", "generic-f32", "generic-x3"])
@pytest.mark.parametrize("name", ["mcc2_cls_h128_L2_B3", "mck4_cls_h128_L2_B3"])
def test_fused_cross_entropy_backward_matches_golden(name, route, monkeypatch):
    """Classification wrappers: CE over the per-foot logit pairs fused into the decoder backward (mshgnn_backward_ce) gives
    the loss and every gradient of the reference run (golden vectors) at 1e-4 on every parity-grade route (both engines)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    if route.startswith("generic-"):
        monkeypatch.setenv("MSHGNN_ENGINE", "generic")
    e = eng.Engine(spec, route.split("-")[-1])
    if route != "x3":      # (the split plan of a 20-node K4 window does not fit the LDS-resident tile: that one runs on the generic engine anyway)
        assert e.generic == route.startswith("generic-")
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    out = e.forward(xs, flat, B, training=True)
    loss, gflat = e.backward_ce(xs, flat, out, y.reshape(B, 4).to(e.device, torch.int32).contiguous(), B)
    torch.cuda.synchronize()
    grads = {k: v.detach().cpu() for k, v in eng.unflatten(spec, gflat).items()}
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().cpu(), grads, rtol=RTOL_F32, what=name)


@pytest.mark.parametrize("dtype,B", [("bf16", 37), ("bf16", 8192), ("f32", 37)])
def test_one_call_step_equals_forward_plus_backward(dtype, B):
    """mshgnn_step_mse (bf16 plan: decoder + MSE + decoder backward inside the fused forward kernel) == forward followed by
    backward_mse: same output bits, loss and gradients up to fp32 summation order."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    e = eng.Engine(spec, dtype)
    x_dict, y = synth.make_windows(21, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    yd = y.reshape(-1).to(e.device, torch.float32)
    flat = eng.flatten_params(spec, synth.make_params(21, spec.param_shapes()), e.device)
    out_a = e.forward(xs, flat, B).clone()
    loss_a, g_a = e.backward_mse(xs, flat, out_a, yd, B)
    loss_a, g_a = loss_a.clone(), g_a.clone()
    out_b, loss_b, g_b = e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-5 * abs(float(loss_a))
    assert float((g_a - g_b).abs().max() / g_a.abs().max()) < 2e-5


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_two_phase_step_is_bit_identical(dtype):
    """mshgnn_step_mse_phase (what bench.py interleaves with the all-reduce on N > 1 GPUs): after phase 0 everything but the
    encoder's gradients is final, after phase 1 the whole buffer equals the one-call step bit for bit."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 333
    e = eng.Engine(spec, dtype)
    x_dict, y = synth.make_windows(5, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    yd = y.reshape(-1).to(e.device, torch.float32)
    flat = eng.flatten_params(spec, synth.make_params(5, spec.param_shapes()), e.device)
    out_a, loss_a, g_a = e.step_mse(xs, flat, yd, B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    split = int(e.info.grad_split)
    offs = spec.param_offsets()
    assert split == offs["convs.0.convs.<base___front_bj___joint>.lin_rel.weight"][0] == min(o for k, (o, n) in offs.items() if not k.startswith("encoder."))
    out_b = torch.empty_like(out_a); loss_b = torch.empty(1, device=e.device); g_b = torch.full_like(g_a, float("nan"))
    e.step_mse_phase(0, xs, flat, yd, B, out_b, g_b, loss_b)
    torch.cuda.synchronize()
    assert torch.equal(g_b[split:], g_a[split:]) and torch.equal(loss_b, loss_a) and torch.equal(out_b, out_a)
    assert bool(torch.isnan(g_b[:split]).all())           # the encoder's slice is untouched until phase 1
    e.step_mse_phase(1, xs, flat, yd, B, out_b, g_b, loss_b)
    torch.cuda.synchronize()
    assert torch.equal(g_b, g_a)


def test_adam_step_matches_torch_adam():
    """mshgnn_adam_step vs torch.optim.Adam (fp64, CPU) over 3 steps of the engine's own gradients (optimizer
    pinned by post-step parameters)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    B = 5
    e = eng.Engine(spec, "f32")
    x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    ref = flat.detach().cpu().double().clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    yd = y.reshape(-1).to(e.device, torch.float32)
    for step in range(1, 4):
        out = e.forward(xs, flat, B)
        _, g = e.backward_mse(xs, flat, out, yd, B)
        ref.grad = g.detach().cpu().double()
        opt.step()
        e.adam_step(flat, g, m, v, step, lr=1e-3)
        torch.cuda.synchronize()
        delta_ref = (ref.detach() - flat.cpu().double()).abs().max()
        assert float(delta_ref) < 2e-6, (step, float(delta_ref))   # updates are O(lr)=1e-3; fp32 state vs fp64 reference


def test_step_is_hip_graph_capturable():
    """forward + fused-loss backward launch only kernels on the caller's stream (no allocation / sync inside the C-ABI),
    so a whole step can be captured in a HIP graph and replayed; replay reproduces the eager gradients bit for bit."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 64
    e = eng.Engine(spec, "bf16")
    x_dict, y = synth.make_windows(4, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, synth.make_params(4, spec.param_shapes()), e.device)
    yd = y.reshape(-1).to(e.device, torch.float32)
    out = torch.empty(B * 4, 3, dtype=torch.float32, device=e.device)
    gflat = torch.empty_like(flat)
    loss = torch.empty(1, device=e.device)

    def step():
        e.forward(xs, flat, B, training=True, out=out)
        e.backward_mse(xs, flat, out, yd, B, grad_flat=gflat, loss=loss)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = gflat.clone()
    ref_loss = float(loss)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    gflat.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(gflat, ref) and float(loss) == ref_loss


@pytest.mark.parametrize("kind,topo,cfg,B", [("c2", "a1-c2", "a1-c2", 8192), ("c2", "a1-c2", "a1-c2", 1000), ("c2", "a1-c2", "a1-c2", 4800), ("mi", "quadruped-mi", "", 530),
                                             ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 8192), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 777),
                                             ("k4_com", "solo-k4-com", "solo-k4", 500), ("c2_com", "solo-c2-com", "solo-c2", 300)])
def test_slab_and_eight_wave_stack_kernels_agree_bit_for_bit(kind, topo, cfg, B, monkeypatch):
    """The slab stack kernels (two 4-wave workgroups per CU, destination nodes in two groups) accumulate every
    node in the same order as the 8-wave stack kernels: outputs and
    every gradient but the decoder's (whose per-tile partials are summed over fewer per-wave partials) are identical bits, for full-size and
    ragged batches -- and for whichever of the two the plan picks by itself ("default": the slab kernels from the first tile beyond one per CU, e.g. the 300
    tiles of 4800 windows)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec(kind, topo, cfg, 128, 3, grf=3 if kind == "c2" else 1)
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(5, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(5, spec.param_shapes())
    res = {}
    # "8wave": the two launches per step of the 8-wave kernels; "8wave-step" and "slab" run both sweeps in one launch (k_stack_step / k_slab_step)
    modes = {"slab": ("2", "1"), "slab-2launch": ("2", "0"), "8wave-step": ("0", "1"), "8wave": ("0", "0"), "default": (None, "1")}
    for mode, (slab, step) in modes.items():
        if slab is None:
            monkeypatch.delenv("MSHGNN_SLAB", raising=False)
        else:
            monkeypatch.setenv("MSHGNN_SLAB", slab)       # read when the plan is created
        monkeypatch.setenv("MSHGNN_STEP_KERNEL", step)
        e = eng.Engine(spec, "bf16")
        if mode.startswith("slab") and not (e.info.kernel_sets & 2):
            continue
        xs = e.cast_inputs(x_dict)
        yd = y.reshape(-1).to(e.device, torch.float32)
        flat = eng.flatten_params(spec, params, e.device)
        out, loss, g = e.step_mse(xs, flat, yd, B)
        torch.cuda.synchronize()
        res[mode] = (out.clone(), loss.clone(), g.clone())
    ref = res["8wave"]
    for mode in res:
        if mode == "8wave":
            continue
        assert torch.equal(res[mode][0], ref[0]), mode
        ga, gb = eng.unflatten(spec, res[mode][2]), eng.unflatten(spec, ref[2])
        for k in ga:
            if k.startswith("decoder") and mode not in ("8wave-step",):     # summed over 4 instead of 8 per-wave partials per tile: fp32 summation order
                assert float((ga[k] - gb[k]).abs().max()) <= 2e-6 * float(gb[k].abs().max()), (mode, k)
            else:
                assert torch.equal(ga[k], gb[k]), (mode, k)
        assert abs(float(res[mode][1]) - float(ref[1])) <= 1e-6 * abs(float(ref[1])), mode


@pytest.mark.parametrize("kind,topo,cfg,layers,B", [("c2", "a1-c2", "a1-c2", 8, 300), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 8, 130)])
def test_slab_stack_kernels_on_every_entry_point(kind, topo, cfg, layers, B, monkeypatch):
    """The slab stack kernels (forced for a small batch, MSHGNN_SLAB=2) behind every entry point that launches a stack kernel -- inference forward (no
    stashes), the two-call route (mshgnn_forward + mshgnn_backward_mse: the decoder backward is its own launch) and, for the classification model, the
    one-call cross-entropy step -- at the paper's depth (L = 8): identical bits to the 8-wave kernels (decoder gradients: summation order)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    reg = kind == "c2"
    spec = helpers.make_spec(kind, topo, cfg, 128, layers, grf=3 if reg else 1, regression=reg)
    n_y = spec.out_channels * spec.num_nodes[spec.out_type] if reg else spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(11, B, spec.num_nodes, spec.widths, n_y, classification=not reg)
    params = synth.make_params(11, spec.param_shapes())
    res = {}
    for mode, slab in {"slab": "2", "8wave": "0"}.items():
        monkeypatch.setenv("MSHGNN_SLAB", slab)
        e = eng.Engine(spec, "bf16")
        assert e.info.kernel_sets & 2
        xs = e.cast_inputs(x_dict)
        flat = eng.flatten_params(spec, params, e.device)
        inf = e.forward(xs, flat, B, training=False).clone()
        if reg:
            yd = y.reshape(-1).to(e.device, torch.float32)
            out = e.forward(xs, flat, B, training=True).clone()
            g = e.backward_mse(xs, flat, out, yd, B)[1].clone()
        else:
            lab = y.reshape(B, -1).to(e.device, torch.int32)
            out, loss, g = e.step_ce(xs, flat, lab, B)
            out, g = out.clone(), g.clone()
        torch.cuda.synchronize()
        res[mode] = (inf, out, g)
    assert torch.equal(res["slab"][0], res["8wave"][0]) and torch.equal(res["slab"][1], res["8wave"][1])
    ga, gb = eng.unflatten(spec, res["slab"][2]), eng.unflatten(spec, res["8wave"][2])
    for k in ga:
        if k.startswith("decoder"):
            assert float((ga[k] - gb[k]).abs().max()) <= 2e-6 * float(gb[k].abs().max()), k
        else:
            assert torch.equal(ga[k], gb[k]), k


@pytest.mark.parametrize("dtype,kind,topo,cfg,layers,B", [("bf16", "c2", "a1-c2", "a1-c2", 3, 8192), ("x3", "c2", "a1-c2", "a1-c2", 3, 1000),
                                                          ("bf16", "k4", "mini_cheetah-k4", "mini_cheetah-k4", 4, 777), ("f32", "c2", "a1-c2", "a1-c2", 2, 130)])
def test_node_level_liveness_changes_no_result(dtype, kind, topo, cfg, layers, B, monkeypatch):
    """The plan computes only the nodes whose values can reach the decoder within the model's depth (spec.node_liveness; A1-C2 at 3 layers: not the
    base nodes).  Against the same plan with every node of every live type computed (MSHGNN_PRUNE=0, the liveness of rounds 1-3) at the headline batch:
    the outputs and the loss are IDENTICAL BITS (a live node's accumulation order is untouched), every gradient agrees to fp32 summation order (the
    weight-gradient launch cuts the batch into another number of window parts), and the parameters the topology predicts dead are exact zeros in both."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    from tests.test_liveness import predicted_dead
    spec = helpers.make_spec(kind, topo, cfg, 128, layers, grf=3 if kind == "c2" else 1)
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(17, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(17, spec.param_shapes())
    res = {}
    for prune in ("1", "0"):
        monkeypatch.setenv("MSHGNN_PRUNE", prune)      # read when the plan is compiled
        e = eng.Engine(spec, dtype)
        out, loss, g = e.step_mse(e.cast_inputs(x_dict), eng.flatten_params(spec, params, e.device), y.reshape(-1).to(e.device, torch.float32), B)
        torch.cuda.synchronize()
        res[prune] = (out.clone(), loss.clone(), eng.unflatten(spec, g.clone()), e.info.flops_fwd)
    monkeypatch.setenv("MSHGNN_PRUNE", "1")
    dead = predicted_dead(spec)
    assert res["1"][3] < res["0"][3] and dead, "the case must have dead nodes inside live types"
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    for k, ga in res["1"][2].items():
        gb = res["0"][2][k]
        if k in dead:
            assert float(ga.abs().max()) == 0.0 and float(gb.abs().max()) == 0.0, k
        else:
            assert float(gb.abs().max()) > 0.0 and float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()), k


def test_full_size_batch_properties_bf16():
    """The same size-independent properties on the throughput plan (bf16, B=8192: slab stack kernels; the two halves of the
    batch run on the 8-wave stack kernels): every window's output is independent of its batch -- identical bits -- and the
    gradient of the halves adds up to the gradient of the whole (fp32 summation order only)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 8192
    e = eng.Engine(spec, "bf16")
    x_dict, y = synth.make_windows(9, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    yd = y.reshape(-1).to(e.device, torch.float32)
    flat = eng.flatten_params(spec, synth.make_params(9, spec.param_shapes()), e.device)
    out, loss, g = e.step_mse(xs, flat, yd, B)
    out, loss, g = out.clone(), loss.clone(), g.clone()
    h = B // 2
    res = []
    for lo in (0, h):
        xh = [x.view(B, -1)[lo:lo + h].reshape(h * spec.num_nodes[t], -1).contiguous() for x, t in zip(xs, spec.node_types)]
        oh, lh, gh = e.step_mse(xh, flat, yd.view(B, -1)[lo:lo + h].reshape(-1).contiguous(), h)
        res.append((oh.clone(), lh.clone(), gh.clone()))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([res[0][0], res[1][0]]), out)
    # each half's loss / gradient is a mean over h windows: the whole batch is their average
    assert abs(0.5 * (float(res[0][1]) + float(res[1][1])) - float(loss)) <= 1e-5 * abs(float(loss))
    assert float((0.5 * (res[0][2] + res[1][2]) - g).abs().max() / g.abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16", "x3"])
@pytest.mark.parametrize("kind,topo,cfg,regression", [("k4", "mini_cheetah-k4", "mini_cheetah-k4", False), ("k4_com", "solo-k4-com", "solo-k4", True),
                                                       ("c2_com", "solo-c2-com", "solo-c2", True)])
def test_k4_and_com_batches_of_a_thousand_windows_are_additive(kind, topo, cfg, regression, dtype):
    """MiniCheetah-K4 classification (BASELINE configs[2]) and the Solo COM graphs (configs[3]) at B = 1000, past the sizes the oracle
    covers: every window's output is independent of its batch (identical bits) and the gradient w.r.t. a fixed dL/dout of the two halves
    adds up to the gradient of the whole -- on the fp32 plan, the bf16 plan and the split plan (MiniCheetah-K4 at 'x3': generic engine)."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec(kind, topo, cfg, 128, 3, regression=regression)
    B = 1000
    n_out, d_out = spec.num_nodes[spec.out_type], spec.out_channels
    e = eng.Engine(spec, dtype)
    x_dict, _ = synth.make_windows(13, B, spec.num_nodes, spec.widths, n_out * d_out)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, synth.make_params(13, spec.param_shapes()), e.device)
    g = torch.Generator().manual_seed(5)
    go = torch.randn(B * n_out, d_out, generator=g).to(e.device)
    out = e.forward(xs, flat, B).clone()
    g_all = e.backward(xs, flat, go, B).clone()
    h = B // 2
    parts = []
    for lo in (0, h):
        xh = [x.view(B, -1)[lo:lo + h].reshape(h * spec.num_nodes[t], -1).contiguous() for x, t in zip(xs, spec.node_types)]
        oh = e.forward(xh, flat, h).clone()
        gh = e.backward(xh, flat, go[lo * n_out:(lo + h) * n_out].contiguous(), h).clone()
        parts.append((oh, gh))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([parts[0][0], parts[1][0]]), out)
    assert float((parts[0][1] + parts[1][1] - g_all).abs().max() / g_all.abs().max()) < (2e-3 if dtype == "bf16" else 1e-4)


def test_engine_keeps_a_bounded_number_of_workspaces():
    """A sweep over batch sizes must not pin one workspace per size forever (0.6 GB each at B = 8192): least recently used ones are released,
    and a forward whose workspace was released can no longer be backpropagated (its ticket changes) instead of reading freed memory."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    e = eng.Engine(spec, "bf16")
    flat = eng.flatten_params(spec, synth.make_params(1, spec.param_shapes()), e.device)
    t0 = None
    for B in (3, 5, 7, 9, 11, 13):
        x_dict, _ = synth.make_windows(B, B, spec.num_nodes, spec.widths, 12)
        e.forward(e.cast_inputs(x_dict), flat, B)
        if B == 3:
            t0 = e.stash_ticket(3)
    torch.cuda.synchronize()
    assert len(e._ws) <= e.MAX_WORKSPACES and (13, 1) in e._ws and (3, 1) not in e._ws
    assert e.stash_ticket(3) != t0


@pytest.mark.parametrize("kind,topo,cfg,L", [("c2", "a1-c2", "a1-c2", 2), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 8)])
def test_bf16_unaligned_inputs_take_the_element_wise_loaders(kind, topo, cfg, L):
    """Dense (unpadded) bf16 inputs: joint rows of 450 / 300 elements start 4-byte aligned only -> the element-wise loaders of the encoder and
    of the weight-gradient kernel (k_gradw_bf16_lean<false>, several items per lane on the K4 plan); same results as the padded layout."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec(kind, topo, cfg, 128, L, grf=3 if kind == "c2" else 1)
    B = 21
    e = eng.Engine(spec, "bf16")
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    yd = y.reshape(-1).to(e.device, torch.float32)
    o1, l1, g1 = e.step_mse(e.cast_inputs(x_dict, pad=True), flat, yd, B)
    o1, l1, g1 = o1.clone(), l1.clone(), g1.clone()
    o2, l2, g2 = e.step_mse(e.cast_inputs(x_dict, pad=False), flat, yd, B)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(g1, g2)


@pytest.mark.parametrize("dtype", ["bf16", "x3", "f32"])
@pytest.mark.parametrize("kind,topo,cfg,B", [("k4", "mini_cheetah-k4", "mini_cheetah-k4", 8192), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 333),
                                             ("c2", "mini_cheetah-c2", "mini_cheetah-c2", 77)])
def test_one_call_classification_step_equals_forward_plus_backward_ce(kind, topo, cfg, B, dtype):
    """mshgnn_step_ce (bf16 plan: decoder + cross entropy + decoder backward in the tail of the fused forward kernel) == mshgnn_forward
    followed by mshgnn_backward_ce: same logits bits, loss and gradients up to fp32 summation order."""
    _require_gpu()
    from morphsym_hgnn_amd import engine as eng, synth
    if dtype != "bf16" and B > 1000:
        pytest.skip("the full-size batch exercises the slab kernels of the bf16 plan")
    spec = helpers.make_spec(kind, topo, cfg, 128, 3, regression=False)
    e = eng.Engine(spec, dtype)
    x_dict, y = synth.make_windows(31, B, spec.num_nodes, spec.widths, 4, classification=True)
    xs = e.cast_inputs(x_dict)
    lab = y.reshape(B, -1).to(e.device, torch.int32).contiguous()
    flat = eng.flatten_params(spec, synth.make_params(31, spec.param_shapes()), e.device)
    out_a = e.forward(xs, flat, B).clone()
    loss_a, g_a = e.backward_ce(xs, flat, out_a, lab, B)
    loss_a, g_a = loss_a.clone(), g_a.clone()
    out_b, loss_b, g_b = e.step_ce(xs, flat, lab, B)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    assert float((g_a - g_b).abs().max()) <= 2e-6 * float(g_a.abs().max())
