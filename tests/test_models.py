"""The drop-in nn.Module surface: same constructor, parameter names, state_dict layout and forward contract as the
reference's GRF_HGNN_C2 / GRF_HGNN_K4 / GRF_HGNN; GPU tests run it against the golden vectors."""
import os

import pytest
import torch
from torch import nn

from morphsym_hgnn_amd import models, nn as pnn
from tests import helpers


def _build(case, spec, **extra):      # extra: e.g. activation_fn
    _, cfg_path = helpers.load_group(case["cfg"])
    meta = spec.topology.metadata()
    if case["kind"] == "c2":
        return models.GRF_HGNN_C2(case["hidden"], case["layers"], meta, regression=case["regression"],
                                  symmetry_mode="MorphSym" if cfg_path else None, group_operator_path=cfg_path,
                                  grf_dimension=case["grf"], **extra)
    if case["kind"] == "k4":
        return models.GRF_HGNN_K4(case["hidden"], case["layers"], meta, regression=case["regression"],
                                  symmetry_mode="MorphSym" if cfg_path else None, group_operator_path=cfg_path, **extra)
    if case["kind"] in ("k4_com", "c2_com"):
        cls = models.COM_HGNN_K4 if case["kind"] == "k4_com" else models.COM_HGNN_C2
        return cls(case["hidden"], case["layers"], meta, symmetry_mode="MorphSym" if cfg_path else None,
                   group_operator_path=cfg_path, **extra)
    if case["kind"] == "s4_com":
        if case.get("ref") == "COM_HGNN":
            return models.COM_HGNN(case["hidden"], case["layers"], meta, com_dimension=6, **extra)
        return models.COM_HGNN_S4(case["hidden"], case["layers"], meta, **extra)
    return models.GRF_HGNN(case["hidden"], case["layers"], meta, regression=case["regression"], grf_dimension=case["grf"], **extra)


def _is_com(case):
    return case["kind"].endswith("_com")


@pytest.mark.parametrize("name", ["a1c2_h128_L3_d3_B3", "mck4_cls_h128_L2_B3", "mi_h128_L2_d1_B3", "solok4com_h128_L3_B5",
                                  "soloc2com_h128_L2_B4", "solos4com_h128_L2_B3", "com_hgnn_h128_L2_B3"])
def test_state_dict_layout_matches_reference_names(name):
    torch.set_default_dtype(torch.float64)   # the reference runs in float64 (gnnLightning.py:1183)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    m = _build(case, spec)
    for t in spec.node_types:   # what the first forward does
        m.encoder.lins[t].materialize(spec.widths[t])
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.param_shapes().keys())
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(spec.param_shapes())
    assert sum(p.numel() for p in m.parameters()) == int(fx["n_params"])
    m.load_state_dict(params)    # golden weights load by name
    et = spec.edge_types[0]
    assert m.convs[0].convs[et].lin_rel.weight.shape == (128, 128)   # tuple-key access, hgnn_c2.py:295-306
    if _is_com(case):
        assert m.num_dimensions_per_base == spec.out_channels == 6 and m.num_bases == spec.num_nodes["base"]
        if case["kind"] != "s4_com":
            c = spec.symmetry_coefficients()
            assert torch.equal(m.joints_linear_weights, c[0]) and torch.equal(m.base_coefficients_lin, c[2])
            assert torch.equal(m.base_coefficients_ang, c[3])
        return
    assert m.out_channels_per_foot == spec.out_channels
    if case["kind"] != "mi":
        c = spec.symmetry_coefficients()
        assert torch.equal(m.joints_linear_weights, c[0]) and torch.equal(m.feet_linear_weights, c[1])


def test_operators_have_no_cpu_path():
    """Called on their own the four operator modules run on the HIP operators of ops.py; host tensors raise (no eager fallback)."""
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pnn.GraphConv(8, 8)(torch.zeros(2, 8), torch.zeros(2, 0, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pnn.Linear(8, 4)(torch.zeros(2, 8))
    m = models.GRF_HGNN_C2(128, 1, (["base", "joint", "foot"], [("base", "x", "joint")]), activation_fn=nn.Tanh())      # accepted: runs operator by operator
    assert not m._fused_activation


@pytest.mark.gpu
@pytest.mark.parametrize("name,param_dev", [("a1c2_h128_L3_d3_B3", "cuda"), ("a1c2_h128_L2_d3_B37", "cpu"),
                                            ("mck4_cls_h128_L2_B3", "cuda"), ("mi_h128_L2_d3_B2", "cuda"),
                                            ("solok4com_h128_L3_B5", "cuda"), ("soloc2com_h128_L2_B4", "cpu"),
                                            ("solos4com_h128_L2_B3", "cuda"), ("com_hgnn_h128_L2_B3", "cuda")])
def test_module_forward_backward_matches_golden(name, param_dev):
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    m = _build(case, spec)
    if param_dev == "cuda":
        m = m.cuda()
    dev = torch.device(param_dev)
    xd = {k: v.to(dev) for k, v in x_dict.items()}
    eid = {k: v.to(dev) for k, v in ei.items()}
    with torch.no_grad():        # lazy init exactly like gnnLightning.py:593-595
        m(x_dict={k: v.clone() for k, v in xd.items()}, edge_index_dict=eid)
    m.load_state_dict(params)
    out = m(x_dict=xd, edge_index_dict=eid)
    if _is_com(case):
        w = m.num_bases * m.num_dimensions_per_base                    # gnnLightning_com.py:335-338
        assert tuple(out.shape) == ((B, m.num_bases, 6) if case["kind"] != "s4_com" else (B, 6))
    else:
        w = m.out_channels_per_foot * 4
    y_pred = torch.reshape(out.squeeze(), (B, w))                      # gnnLightning.py:691
    if case["regression"]:
        loss = ((y_pred.flatten() - y.to(dev).reshape(B, w).flatten()) ** 2).mean()
    else:
        loss = torch.nn.functional.cross_entropy(y_pred.reshape(B * 4, 2), y.to(dev).reshape(B, 4).long().flatten())
    loss.backward()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()}
    assert out.dtype == torch.float64 and out.device.type == dev.type
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().cpu(), grads, rtol=1e-4, what=name)
    # a second step works (fresh flat buffer, stash ticket advances), and wrong topologies are rejected
    m.zero_grad()
    m(x_dict=xd, edge_index_dict=eid).sum().backward()
    bad = dict(eid)
    k0 = spec.edge_types[0]
    bad[k0] = eid[k0].flip(1)
    m2 = _build(case, spec)
    if B > 1:
        with pytest.raises(ValueError):
            m._checked_batches.clear()
            m(x_dict=xd, edge_index_dict=bad)


@pytest.mark.gpu
def test_device_parameters_are_views_of_the_flat_buffer_and_behave_like_ordinary_parameters():
    """Fast path of the module surface: with the module on the GPU every parameter is an fp32 view into the engine's flat buffer (nothing
    is copied per forward), .grad tensors are views of the flat gradient.  What a training loop relies on must keep working:
    gradient accumulation over two backwards, an in-place optimizer step seen by the next forward, load_state_dict, and
    .double() / .to() followed by another forward."""
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    B = case["B"]
    m = _build(case, spec).cuda()
    xd = {k: v.cuda() for k, v in x_dict.items()}
    eid = {k: v.cuda() for k, v in ei.items()}
    with torch.no_grad():
        m(x_dict=dict(xd), edge_index_dict=eid)
    m.load_state_dict(params)
    flat = m._flat
    assert all(p.dtype == torch.float32 and p.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in m.parameters())

    def loss_of(model):
        out = model(x_dict=dict(xd), edge_index_dict=eid)
        return ((out.flatten() - y.cuda().flatten()) ** 2).mean()

    loss_of(m).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    helpers.check_against_fixture(fx, m(x_dict=dict(xd), edge_index_dict=eid).detach().cpu(), None, {k: v.cpu() for k, v in g1.items()} | {
        k: torch.zeros_like(p).cpu() for k, p in m.named_parameters() if p.grad is None}, rtol=1e-4, what="module fast path")
    loss_of(m).backward()                                    # accumulation: .grad == 2 x the single-step gradient
    for k, p in m.named_parameters():
        if k in g1:
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-6, atol=0), k
    m.zero_grad(set_to_none=True)
    l0 = loss_of(m)
    l0.backward()
    l0 = float(l0)
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    opt.step()                                               # in place on the views -> the flat buffer the kernels read
    assert m._flat is flat and float(loss_of(m)) < l0
    m.double()                                               # replaces the parameter tensors ...
    assert not m._flat_ok
    l1 = float(loss_of(m))                                   # ... the next forward re-establishes the fp32 views
    assert m._flat_ok and all(p.dtype == torch.float32 for p in m.parameters()) and abs(l1 - float(loss_of(m))) < 1e-12


@pytest.mark.gpu
def test_gradient_accumulation_with_frozen_parameters_and_after_to():
    """Gradient delivery decides per TRAINABLE parameter (the first parameters in flat order may be frozen: their .grad is always None), and a
    .grad tensor that is not a view of the current flat gradient buffer (left over from before .to() re-created the buffers, gradients zeroed
    in place) is accumulated into instead of being dropped."""
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    xd = {k: v.cuda() for k, v in x_dict.items()}
    eid = {k: v.cuda() for k, v in ei.items()}

    def loss_of(model):
        out = model(x_dict=dict(xd), edge_index_dict=eid)
        return ((out.flatten() - y.cuda().flatten()) ** 2).mean()

    def fresh():
        m = _build(case, spec).cuda()
        with torch.no_grad():
            m(x_dict=dict(xd), edge_index_dict=eid)
        m.load_state_dict(params)
        return m
    ref = fresh()
    loss_of(ref).backward()
    g1 = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    # (a) frozen encoder, two accumulation steps
    m = fresh()
    for k, p in m.named_parameters():
        if k.startswith("encoder"):
            p.requires_grad_(False)
    loss_of(m).backward()
    loss_of(m).backward()
    for k, p in m.named_parameters():
        if k.startswith("encoder"):
            assert p.grad is None, k
        elif k in g1:
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-6, atol=0), k
    # (b) .to() between two steps with the gradients zeroed in place: the old .grad tensors survive and must receive the new gradient
    m = fresh()
    loss_of(m).backward()
    m.zero_grad(set_to_none=False)
    held = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    m.double()                                     # replaces the parameter tensors (and keeps their .grad), the next forward rebuilds the flat buffers
    loss_of(m).backward()
    for k, p in m.named_parameters():
        if k in g1:
            assert p.grad is not None and torch.allclose(p.grad.float(), g1[k], rtol=1e-5, atol=0), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a1c2_h128_L3_d3_B3", "mck4_cls_h128_L2_B3", "mi_h128_L2_d3_B2", "solok4com_h128_L3_B5", "solos4com_h128_L2_B3"])
def test_non_relu_activation_runs_operator_by_operator(name):
    """activation_fn other than nn.ReLU() (the reference's constructors take any module): the same forward on the stand-alone HIP operators;
    outputs, loss and every parameter gradient within 1e-4 of the oracle evaluated with that activation (base_transform keeps its own nn.ReLU,
    hgnn_c2.py:117-121)."""
    from oracle import ms_hgnn_oracle as orc
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    m = _build(case, spec, activation_fn=nn.Tanh()).cuda()
    dev = torch.device("cuda")
    xd = {k: v.to(dev) for k, v in x_dict.items()}
    eid = {k: v.to(dev) for k, v in ei.items()}
    with torch.no_grad():
        m(x_dict={k: v.clone() for k, v in xd.items()}, edge_index_dict=eid)      # lazy encoder, spec
    m.load_state_dict(params)
    out = m(x_dict=xd, edge_index_dict=eid)
    # oracle with tanh at the encoder / layer sites and relu inside base_transform
    cfg = helpers.oracle_config(spec)
    act = lambda key, h: torch.relu(h) if key[0] == "t1" else torch.tanh(h)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    o_ref = orc.forward(cfg, leaves, {k: v.clone() for k, v in x_dict.items()}, ei, relu_fn=act)
    yy, yp = orc.wrapper_outputs(cfg, o_ref, y, B)
    l_ref = orc.mse_loss(yy, yp) if cfg.regression else orc.cross_entropy_loss(yy, yp, B)
    l_ref.backward()
    yy2, yp2 = orc.wrapper_outputs(cfg, out.cpu(), y, B)
    assert float((yp2 - yp.detach()).abs().max() / yp.detach().abs().max()) < 1e-4
    loss = (orc.mse_loss(yy.to(dev), orc.wrapper_outputs(cfg, out, y.to(dev), B)[1]) if cfg.regression
            else orc.cross_entropy_loss(yy.to(dev), orc.wrapper_outputs(cfg, out, y.to(dev), B)[1], B))
    loss.backward()
    assert abs(float(loss) - float(l_ref)) <= 1e-4 * abs(float(l_ref))
    for k, p in m.named_parameters():
        ref = leaves[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert float((p.grad.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,precision,kind", [(96, "f32", "c2"), (96, "x3", "c2"), (64, "x3", "k4"), (192, "x3", "c2"), (200, "bf16", "c2")])
def test_hidden_width_off_the_engines_grid_runs_fused_through_zero_padding(hidden, precision, kind, monkeypatch):
    """hidden_channels that is not a multiple of 128 (the reference takes any width, hgnn_c2.py:11; the fused engines take multiples of 128): the
    model runs on the fused engine of the next multiple with zero rows / columns (engine.PaddedEngine: 96 / 64 -> the LDS-resident kernels at 128,
    192 / 200 -> the generic-width engine at 256) and gives the oracle's output, loss and every gradient at the plan's tolerance; the caller's
    parameters, gradients and state_dict keep their true shapes."""
    from oracle import ms_hgnn_oracle as orc
    from morphsym_hgnn_amd import engine as eng, synth
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    monkeypatch.setenv("MSHGNN_DTYPE", precision)
    topo = "a1-c2" if kind == "c2" else "mini_cheetah-k4"
    spec = helpers.make_spec(kind, topo, topo, hidden, 2)
    B = 5
    _, cfg_path = helpers.load_group(topo)
    if kind == "c2":
        m = models.GRF_HGNN_C2(hidden, 2, spec.topology.metadata(), symmetry_mode="MorphSym", group_operator_path=cfg_path, grf_dimension=3).cuda()
    else:
        m = models.GRF_HGNN_K4(hidden, 2, spec.topology.metadata(), symmetry_mode="MorphSym", group_operator_path=cfg_path).cuda()
    assert m._fused_activation
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(4, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(4, spec.param_shapes())
    ei = spec.topology.edge_index_dict(B)
    xd = {k: v.double().cuda() for k, v in x_dict.items()}
    eid = {k: v.cuda() for k, v in ei.items()}
    with torch.no_grad():
        m(x_dict={k: v.clone() for k, v in xd.items()}, edge_index_dict=eid)
    m.load_state_dict(params)
    out = m(x_dict=xd, edge_index_dict=eid)
    e = next(iter(m._engines.values()))
    assert isinstance(e, eng.PaddedEngine) and e.inner_spec.hidden == (hidden + 127) // 128 * 128 and e.generic == (hidden > 128)
    assert all(tuple(p.shape) == tuple(spec.param_shapes()[k]) for k, p in m.named_parameters())
    tol_out, tol_grad = (1e-4, 1e-4) if precision != "bf16" else (2e-2, 8e-2)
    o_ref, l_ref, g_ref = orc.step(helpers.oracle_config(spec), params, {k: v.double() for k, v in x_dict.items()}, ei, y.double(), B)
    assert float((out.cpu().reshape(-1) - o_ref.reshape(-1)).abs().max() / o_ref.abs().max()) < tol_out
    loss = ((out.flatten() - y.double().cuda().flatten()) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(l_ref)) <= tol_out * abs(float(l_ref))
    for k, p in m.named_parameters():
        ref = g_ref[k]
        if float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        elif precision != "bf16":
            assert float((p.grad.cpu().double() - ref).abs().max() / ref.abs().max()) < tol_grad, k
        else:
            assert float((p.grad.cpu().double() - ref).norm() / ref.norm()) < tol_grad, k


def test_pack_helpers_follow_the_reference_layout():
    """unpack_data / pack_data / ms_foot_decoder (hgnn_c2.py:184-189, 233-284): node rows are [variable][axis][time]; unpacked tensors are
    [batch, time, node * axis]; the pair is an exact round trip and ms_foot_decoder applies the feet's mask to [B, 12]."""
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    m = _build(case, spec)
    B, n, T, D = 3, 2, m.num_timesteps, 3
    data = torch.arange(B * n * 2 * D * T, dtype=torch.float64).view(B * n, 2 * D * T)
    f_p, f_v = m.unpack_data(data, B, n)
    assert f_p.shape == (B, T, n * D) and f_v.shape == (B, T, n * D)
    for b, node, axis, t in ((0, 0, 0, 0), (2, 1, 2, 149), (1, 0, 1, 77)):
        assert f_p[b, t, node * D + axis] == data[b * n + node, axis * T + t]
        assert f_v[b, t, node * D + axis] == data[b * n + node, D * T + axis * T + t]
    assert torch.equal(m.pack_data(f_p, f_v, B, n), data)
    dec = torch.randn(B * 4, 3)
    assert torch.equal(m.ms_foot_decoder(dec), dec.view(B, 12) * m.feet_linear_weights)
    # apply_symmetry == unpack, scale, pack for the base rows (hgnn_c2.py:220-229)
    base = torch.randn(B * 2, 900)
    lin, ang = m.unpack_data(base, B, 2)
    ref = m.pack_data(lin * m.base_coefficients_lin.view(1, 1, -1), ang * m.base_coefficients_ang.view(1, 1, -1), B, 2)
    m._spec = spec
    got = m.apply_symmetry({"base": base.clone(), "joint": torch.randn(B * 12, 450), "foot": torch.ones(B * 4, 1)})["base"]
    assert torch.equal(got, ref)
