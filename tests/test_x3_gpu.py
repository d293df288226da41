"""GPU parity tests of the split-bf16 parity plan (MSHGNN_BF16X3, "x3"): bf16 MFMA with hi/lo operands at the north_star's
1e-4 relative tolerance -- same harness and tolerance as the fp32 plan (tests/test_engine_gpu.py)."""
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
RTOL = 1e-4

# every golden topology at h = 128: the doubled LDS tile holds 2 x (nodes + base_transform nodes) <= 40 blocks, or 2 x nodes <= 40 with the
# base_transform scratch aliased onto the last nodes' blocks (MiniCheetah-K4: 2 x 20)
X3_CASES = [c for c in helpers.GOLDEN_CASES if "_h128_" in c]


@pytest.mark.parametrize("name", X3_CASES)
def test_x3_plan_matches_oracle_and_golden(name):
    assert torch.cuda.is_available()
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, f"{name}: stages above {RTOL}: {bad}"
    helpers.check_against_fixture(fx, out, loss if spec.regression else None, grads, rtol=RTOL, what=name)


def test_x3_k4_runs_on_the_stack_kernels_with_aliased_scratch_and_wide_plans_fall_to_the_generic_engine():
    """MiniCheetah-K4: 2 x 20 node blocks fill the LDS, the base_transform scratch aliases the foot nodes' blocks (k_stack_fwd_x3<true>); a
    request the LDS-resident kernels cannot hold at all (h = 256) runs on the generic-width engine's split arithmetic."""
    from morphsym_hgnn_amd import engine as eng
    case, spec, *_ = helpers.load_case("mck4_cls_h128_L2_B3")
    e = eng.Engine(spec, "x3")
    assert not e.generic and e.storage == "x3" and (e.info.kernel_sets & 1)
    wide = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 256, 2, regression=False)
    e2 = eng.Engine(wide, "x3")
    assert e2.generic and e2.storage == "x3"


@pytest.mark.parametrize("B", [2, 17, 100, 1000])
def test_x3_k4_ragged_batches(B):
    """(B = 1 is left out: with one output channel the decoder bias gradient is the sum of four residuals, a single number that may
    cancel to 1e-3 of its terms -- its RELATIVE error then says nothing about the kernel.)"""
    from morphsym_hgnn_amd import synth
    spec = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 3, grf=1)
    params = synth.make_params(6, spec.param_shapes())
    x_dict, y = synth.make_windows(200 + B, B, spec.num_nodes, spec.widths, 4)
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, (B, bad)


@pytest.mark.parametrize("B", [1, 15, 16, 17, 33, 65, 333])
def test_x3_ragged_batches(B):
    """Batch sizes around the 16-window tile and the 64-window encoder / weight-gradient steps, against the oracle (evaluated with the
    engine's relu decisions, helpers.run_engine_case: a pre-activation within rounding error of zero may be decided either way)."""
    from morphsym_hgnn_amd import synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    params = synth.make_params(5, spec.param_shapes())
    x_dict, y = synth.make_windows(100 + B, B, spec.num_nodes, spec.widths, 12)
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="x3")
    bad = {k: v for k, v in errs.items() if v > RTOL}
    assert not bad, (B, bad)


def test_x3_one_call_step_equals_forward_plus_backward_and_repeats():
    """mshgnn_step_mse on the split plan (decoder + MSE + decoder backward in the tail of its fused forward kernel) == forward followed
    by backward_mse: same output bits, loss and gradients up to fp32 summation order; two runs are bit-identical."""
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    for B in (37, 1000):
        e = eng.Engine(spec, "x3")
        x_dict, y = synth.make_windows(21, B, spec.num_nodes, spec.widths, 12)
        xs = e.cast_inputs(x_dict)
        yd = y.reshape(-1).to(e.device, torch.float32)
        flat = eng.flatten_params(spec, synth.make_params(21, spec.param_shapes()), e.device)
        out_a = e.forward(xs, flat, B).clone()
        loss_a, g_a = e.backward_mse(xs, flat, out_a, yd, B)
        loss_a, g_a = loss_a.clone(), g_a.clone()
        out_b, loss_b, g_b = e.step_mse(xs, flat, yd, B)
        out_b, loss_b, g_b = out_b.clone(), loss_b.clone(), g_b.clone()
        out_c, loss_c, g_c = e.step_mse(xs, flat, yd, B)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b) and torch.equal(out_b, out_c) and torch.equal(g_b, g_c) and torch.equal(loss_b, loss_c)
        assert abs(float(loss_a) - float(loss_b)) <= 1e-5 * abs(float(loss_a))
        assert float((g_a - g_b).abs().max() / g_a.abs().max()) < 2e-5


@pytest.mark.parametrize("kind,topo,cfg,B", [("c2", "a1-c2", "a1-c2", 1000), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 333),
                                             ("k4_com", "solo-k4-com", "solo-k4", 500), ("c2_com", "solo-c2-com", "solo-c2", 77)])
def test_x3_step_kernel_and_two_launches_agree_bit_for_bit(kind, topo, cfg, B, monkeypatch):
    """k_stack_step_x3 (forward + backward sweep in one launch: dX_L handed over through LDS as split hi / lo rows, no tile reload) against the two
    launches k_stack_fwd_x3 + k_stack_bwd_x3 (MSHGNN_STEP_KERNEL=0, read per plan): same code in the same order, so output, loss and every gradient
    are identical bits -- A1-C2, MiniCheetah-K4 (the instantiation whose scratch blocks alias the feet's blocks) and the Solo centroidal-momentum models, whose
    out type (the base) comes first: the tail's reduction scratch then sits behind the base blocks (StackArgs.red_off)."""
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec(kind, topo, cfg, 128, 3, grf=3 if kind == "c2" else 1)
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(31, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(31, spec.param_shapes())
    res = {}
    for step in ("1", "0"):
        monkeypatch.setenv("MSHGNN_STEP_KERNEL", step)      # read when the plan is created
        e = eng.Engine(spec, "x3")
        assert not e.generic
        xs = e.cast_inputs(x_dict)
        out, loss, g = e.step_mse(xs, eng.flatten_params(spec, params, e.device), y.reshape(-1).to(e.device, torch.float32), B)
        torch.cuda.synchronize()
        res[step] = (out.clone(), loss.clone(), g.clone())
    for i in range(3):
        assert torch.equal(res["1"][i], res["0"][i]), i


def test_x3_unaligned_inputs_take_the_general_loader():
    """Dense (unpadded) fp32 inputs: joint rows of 450 floats start 8-byte aligned only -> the element-wise loaders of the encoder and
    the weight-gradient kernel; same results as the padded layout."""
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    B = 21
    e = eng.Engine(spec, "x3")
    x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, 12)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    yd = y.reshape(-1).to(e.device, torch.float32)
    o1, l1, g1 = e.step_mse(e.cast_inputs(x_dict, pad=True), flat, yd, B)
    o1, l1, g1 = o1.clone(), l1.clone(), g1.clone()
    o2, l2, g2 = e.step_mse(e.cast_inputs(x_dict, pad=False), flat, yd, B)
    torch.cuda.synchronize()
    # the aligned layout runs the lean weight-gradient kernel (k_gradw_x3_lean), the dense one the general kernel: same matrix slabs, the bias
    # partial sums add the same values in another order
    assert torch.equal(o1, o2)
    assert int((g1 != g2).sum()) <= 64 * 128 and float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max())


def test_x3_two_phase_step_is_bit_identical():
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 333
    e = eng.Engine(spec, "x3")
    x_dict, y = synth.make_windows(5, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict)
    yd = y.reshape(-1).to(e.device, torch.float32)
    flat = eng.flatten_params(spec, synth.make_params(5, spec.param_shapes()), e.device)
    out_a, loss_a, g_a = e.step_mse(xs, flat, yd, B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    split = int(e.info.grad_split)
    out_b = torch.empty_like(out_a); loss_b = torch.empty(1, device=e.device); g_b = torch.full_like(g_a, float("nan"))
    e.step_mse_phase(0, xs, flat, yd, B, out_b, g_b, loss_b)
    torch.cuda.synchronize()
    assert torch.equal(g_b[split:], g_a[split:]) and torch.equal(loss_b, loss_a) and torch.equal(out_b, out_a)
    e.step_mse_phase(1, xs, flat, yd, B, out_b, g_b, loss_b)
    torch.cuda.synchronize()
    assert torch.equal(g_b, g_a)


def test_x3_full_size_batch_properties():
    """BASELINE config size (B=8192): every window's output is independent of its batch (identical bits) and the gradient of the two
    halves adds up to the gradient of the whole (fp32 summation order only)."""
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 8192
    e = eng.Engine(spec, "x3")
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
    assert abs(0.5 * (float(res[0][1]) + float(res[1][1])) - float(loss)) <= 1e-5 * abs(float(loss))
    assert float((0.5 * (res[0][2] + res[1][2]) - g).abs().max() / g.abs().max()) < 1e-4


@pytest.mark.parametrize("name", ["mcc2_cls_h128_L2_B3"])
def test_x3_fused_cross_entropy_backward_matches_golden(name):
    from morphsym_hgnn_amd import engine as eng
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    e = eng.Engine(spec, "x3")
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    out = e.forward(xs, flat, B, training=True)
    loss, gflat = e.backward_ce(xs, flat, out, y.reshape(B, 4).to(e.device, torch.int32).contiguous(), B)
    torch.cuda.synchronize()
    grads = {k: v.detach().cpu() for k, v in eng.unflatten(spec, gflat).items()}
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().cpu(), grads, rtol=RTOL, what=name)
