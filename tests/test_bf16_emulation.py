import pytest
import torch

from oracle import ms_hgnn_oracle as orc
from tests import helpers
from tests.bf16_emulation import emulate_step


@pytest.mark.parametrize("name", ["a1c2_h128_L3_d3_B3", "mck4_cls_h128_L2_B3", "mi_h128_L2_d3_B2", "mcc2_cls_h128_L2_B3"])
def test_emulation_without_rounding_is_the_oracle(name):
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    o_out, o_loss, o_grads = orc.step(helpers.oracle_config(spec), params, x_dict, ei, y, B)
    out, loss, grads = emulate_step(spec, params, x_dict, y, B, quant=False)
    assert float((out.reshape(-1) - o_out.reshape(-1)).abs().max() / o_out.abs().max()) < 1e-12
    assert abs(float(loss - o_loss)) / abs(float(o_loss)) < 1e-12
    for k, g in o_grads.items():
        m = float(g.abs().max())
        if m == 0:
            assert float(grads[k].abs().max()) == 0
        else:
            assert float((grads[k] - g).abs().max()) / m < 1e-11, k


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["slab", "stack", "layers"])
@pytest.mark.parametrize("name", ["a1c2_h128_L2_d3_B37", "a1c2_h128_L3_d3_B3", "mck4_reg_h128_L1_B2", "mi_h128_L2_d1_B3",
                                  "solok4com_h128_L3_B5", "soloc2com_h128_L2_B4", "solos4com_h128_L2_B3", "a1c2_h128_L8_d3_B2"])
def test_bf16_plan_matches_rounding_point_emulation(name, fused, monkeypatch):
    """bf16 plan vs the fp64 model with bf16 rounding at the engine's storage points.  Remaining differences: fp32
    accumulation and rare 1-ulp bf16 re-roundings (2^-8 relative on single elements), hence norm-wise tolerances:
    outputs 4e-3 (max-abs relative), gradients 1.5e-2 (L2 relative).  All three kernel sets of the bf16 plan are covered
    (the switches are read when the plan is created): the slab stack kernels (default where the plan allows them: two 4-wave
    workgroups per CU), the 8-wave stack kernels (MSHGNN_SLAB=0) and the per-layer kernels (MSHGNN_FUSED=0)."""
    assert torch.cuda.is_available()
    from morphsym_hgnn_amd import engine as eng
    monkeypatch.setenv("MSHGNN_FUSED", "0" if fused == "layers" else "1")
    monkeypatch.setenv("MSHGNN_SLAB", "2" if fused == "slab" else "0")     # 2: also for these small batches
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    e = eng.Engine(spec, "bf16")
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    out_inf = e.forward(xs, flat, B, training=False).clone()      # inference: no stashes written
    out = e.forward(xs, flat, B)
    assert torch.equal(out, out_inf)
    loss, g = e.mse_loss(out.view(-1), y.reshape(-1).to(e.device, torch.float32))
    gflat = e.backward(xs, flat, g, B)
    torch.cuda.synchronize()
    grads = {k: v.cpu().double() for k, v in eng.unflatten(spec, gflat).items()}
    r_out, r_loss, r_grads = emulate_step(spec, params, x_dict, y, B, quant=True)
    err_out = float((out.cpu().double().reshape(-1) - r_out.reshape(-1)).abs().max() / r_out.abs().max())
    assert err_out < 4e-3, err_out
    worst = {}
    for k, ref in r_grads.items():
        n = float(ref.norm())
        if n == 0:
            assert float(grads[k].abs().max()) == 0.0, k
            continue
        worst[k] = float((grads[k] - ref).norm()) / n
    bad = {k: v for k, v in worst.items() if v > 1.5e-2}
    assert not bad, bad
    # and a sanity bound against the exact (un-rounded) oracle: bf16 forward within 2e-2 of fp64
    o_out, _, _ = orc.step(helpers.oracle_config(spec), params, x_dict, ei, y, B)
    assert float((out.cpu().double().reshape(-1) - o_out.reshape(-1)).abs().max() / o_out.abs().max()) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["default", "layers"])
@pytest.mark.parametrize("name", ["mck4_cls_h128_L2_B3", "mcc2_cls_h128_L2_B3", "mck4_cls_h128_L8_B2"])
def test_bf16_plan_classification_through_the_fused_cross_entropy(name, fused, monkeypatch):
    """The classification wrappers on the bf16 plan: logits, then mshgnn_backward_ce (cross entropy fused into the decoder backward),
    against the rounding-point emulation with the same loss -- MiniCheetah K4 (per-layer kernels: 24 LDS blocks do not fit the stack
    kernels' slab variant) and C2, at the paper's depth too."""
    assert torch.cuda.is_available()
    from morphsym_hgnn_amd import engine as eng
    monkeypatch.setenv("MSHGNN_FUSED", "0" if fused == "layers" else "1")
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    e = eng.Engine(spec, "bf16")
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    out = e.forward(xs, flat, B)
    loss, gflat = e.backward_ce(xs, flat, out, y.reshape(B, 4).to(e.device, torch.int32).contiguous(), B)
    torch.cuda.synchronize()
    grads = {k: v.cpu().double() for k, v in eng.unflatten(spec, gflat).items()}
    r_out, r_loss, r_grads = emulate_step(spec, params, x_dict, y, B, quant=True)
    assert float((out.cpu().double().reshape(-1) - r_out.reshape(-1)).abs().max() / r_out.abs().max()) < 4e-3
    assert abs(float(loss) - float(r_loss)) / abs(float(r_loss)) < 4e-3
    bad = {}
    for k, ref in r_grads.items():
        n = float(ref.norm())
        if n == 0:
            assert float(grads[k].abs().max()) == 0.0, k
        elif float((grads[k] - ref).norm()) / n > 1.5e-2:
            bad[k] = float((grads[k] - ref).norm()) / n
    assert not bad, bad
