"""The reference's own input tensors, uncast (mshgnn_forward_src / mshgnn_step_mse_src / mshgnn_step_ce_src, engine.WideInputs): fp64 -- the reference's default
dtype, gnnLightning.py:1183 -- or fp32 device tensors at the dense pitch are converted by the encoder in registers, which also writes the plan-dtype rows
the weight-gradient pass reads.  The conversions are the ones torch's .to() makes (round to nearest even, fp64 -> fp32 -> bf16), so the route is
BIT-IDENTICAL to cast-then-step on every output, loss, gradient and materialised row; against the oracle it therefore inherits the cast route's parity
(tests/test_x3_gpu.py, tests/test_bf16_emulation.py), which is re-checked here on the golden vectors for the parity plan."""
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = ["a1c2_h128_L3_d3_B3", "a1c2_h128_L2_d3_B37", "mck4_cls_h128_L2_B3", "mi_h128_L2_d3_B2", "solok4com_h128_L3_B5", "a1c2_h128_L8_d3_B2"]


def _live_rows_equal(spec, e, rows_a, rows_b, B):
    _, need = spec.node_liveness()
    for t, a, b in zip(e.types, rows_a, rows_b):
        F, n = spec.widths[t], spec.num_nodes[t]
        idx = torch.tensor(need[0][t], dtype=torch.long, device=a.device)
        if idx.numel():
            assert torch.equal(a.view(B, n, -1)[:, idx, :F], b.view(B, n, -1)[:, idx, :F]), t


@pytest.mark.parametrize("src", [torch.float64, torch.float32])
@pytest.mark.parametrize("dtype", ["bf16", "x3"])
@pytest.mark.parametrize("name", CASES)
def test_uncast_inputs_are_bit_identical_to_cast_then_step(name, dtype, src):
    from morphsym_hgnn_amd import engine as eng
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    e = eng.Engine(spec, dtype)
    flat = eng.flatten_params(spec, params, e.device)
    x_src = {k: v.to(src) for k, v in x_dict.items()}
    xs_cast = e.cast_inputs(x_src)                                         # host tensors: the cast + re-pitch pass (torch rounds fp64 -> fp32 -> bf16, and so does the encoder)
    assert not isinstance(xs_cast, eng.WideInputs)
    xw = e.cast_inputs({k: v.to(e.device) for k, v in x_src.items()})      # device tensors at the reference's pitch: no cast pass
    if dtype == "x3" and src == torch.float32 and all((spec.widths[t] * 4) % 16 == 0 for t in e.types):
        assert not isinstance(xw, eng.WideInputs)                         # fp32 rows of whole 16-byte chunks go in as they are
        return
    assert isinstance(xw, eng.WideInputs) and xw.pending and xw.src_bytes == (8 if src == torch.float64 else 4)
    n_out = spec.num_nodes[spec.out_type]
    if spec.regression:
        yd = y.reshape(-1).to(e.device, torch.float32)
        ref = [t.clone() for t in e.step_mse(xs_cast, flat, yd, B)]
        got = [t.clone() for t in e.step_mse(xw, flat, yd, B)]
    else:
        yd = y.reshape(B, n_out).to(e.device, torch.int32).contiguous()
        ref = [t.clone() for t in e.step_ce(xs_cast, flat, yd, B)]
        got = [t.clone() for t in e.step_ce(xw, flat, yd, B)]
    torch.cuda.synchronize()
    assert not xw.pending
    for a, b, what in zip(ref, got, ("out", "loss", "grad")):
        assert torch.equal(a, b), (name, dtype, what, float((a - b).abs().max()))
    _live_rows_equal(spec, e, list(xw), xs_cast, B)
    # the two-call route: forward converts and materialises, backward reads the rows
    xw2 = e.cast_inputs({k: v.to(e.device) for k, v in x_src.items()})
    out_a = e.forward(xs_cast, flat, B).clone()
    go = torch.randn(out_a.shape, generator=torch.Generator().manual_seed(1)).to(e.device)
    g_a = e.backward(xs_cast, flat, go, B).clone()
    out_b = e.forward(list(xw2), flat, B).clone()          # (unpacked, as an autograd Function's *xs: recognised by the row buffers' addresses)
    assert not xw2.pending
    g_b = e.backward(xw2, flat, go, B).clone()
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(g_a, g_b)
    # inference forward on a fresh object (no stash): same output
    xw3 = e.cast_inputs({k: v.to(e.device) for k, v in x_src.items()})
    assert torch.equal(e.forward(xw3, flat, B, training=False), out_a)


@pytest.mark.parametrize("dtype", ["bf16", "x3"])
@pytest.mark.parametrize("B", [1, 15, 17, 64, 65, 333])
def test_uncast_inputs_ragged_batches(B, dtype):
    """Batch sizes around the encoder's 64-window workgroups: rows past the batch are never read from the caller's tensor (its last row ends the
    allocation) and never written."""
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    e = eng.Engine(spec, dtype)
    x_dict, y = synth.make_windows(300 + B, B, spec.num_nodes, spec.widths, 12)
    flat = eng.flatten_params(spec, synth.make_params(4, spec.param_shapes()), e.device)
    yd = y.reshape(-1).to(e.device, torch.float32)
    ref = [t.clone() for t in e.step_mse(e.cast_inputs(x_dict), flat, yd, B)]
    xw = e.cast_inputs({k: v.to(e.device) for k, v in x_dict.items()})
    assert isinstance(xw, eng.WideInputs)
    got = [t.clone() for t in e.step_mse(xw, flat, yd, B)]
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["a1c2_h128_L3_d3_B3", "mck4_cls_h128_L2_B3", "solok4com_h128_L3_B5"])
def test_module_surface_with_reference_dtype_inputs_matches_golden_on_the_parity_plan(name, monkeypatch):
    """The nn.Module surface the reference's scripts call, with fp64 device inputs as its datasets produce them, on the parity plan: output, loss and every
    gradient of the reference run (golden vectors) at 1e-4 -- through the uncast route (the engine hands out WideInputs for these tensors)."""
    from morphsym_hgnn_amd import engine as eng
    from tests.test_models import _build      # the reference-style constructors
    monkeypatch.setenv("MSHGNN_DTYPE", "x3")
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    dev = torch.device("cuda", 0)
    torch.set_default_dtype(torch.float64)
    m = _build(case, spec).to(dev)
    xd = {k: v.to(dev) for k, v in x_dict.items()}
    eid = {k: v.to(dev) for k, v in ei.items()}
    with torch.no_grad():        # lazy init exactly like gnnLightning.py:593-595
        m(x_dict={k: v.clone() for k, v in xd.items()}, edge_index_dict=eid)
    m.load_state_dict(params)
    seen = []
    orig = eng.Engine.cast_inputs
    monkeypatch.setattr(eng.Engine, "cast_inputs", lambda self, xdict, pad=True: seen.append(orig(self, xdict, pad)) or seen[-1])
    out = m(x_dict=xd, edge_index_dict=eid)
    assert seen and isinstance(seen[-1], eng.WideInputs) and not seen[-1].pending
    w = out.numel() // B
    y_pred = torch.reshape(out.squeeze(), (B, w))                      # gnnLightning.py:691
    if case["regression"]:
        loss = ((y_pred.flatten() - y.to(dev).reshape(B, w).flatten()) ** 2).mean()
    else:
        loss = torch.nn.functional.cross_entropy(y_pred.reshape(B * 4, 2), y.to(dev).reshape(B, 4).long().flatten())
    loss.backward()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()}
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().cpu(), grads, rtol=1e-4, what=name)


def test_uncast_entry_points_refuse_what_they_cannot_run():
    """The _src entry points run on the bf16 and split plans of the LDS-resident kernels: the fp32 plan and the generic-width engine answer
    MSHGNN_EUNSUPPORTED (the binding casts there: Engine.cast_inputs hands out plain tensors), a bad element size MSHGNN_EINVAL -- before anything is launched."""
    import ctypes as C
    from morphsym_hgnn_amd import engine as eng, synth
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    B = 4
    x_dict, y = synth.make_windows(1, B, spec.num_nodes, spec.widths, 12)
    xd = {k: v.cuda() for k, v in x_dict.items()}
    for dtype, hidden in (("f32", 128), ("bf16", 256)):
        sp = helpers.make_spec("c2", "a1-c2", "a1-c2", hidden, 2)
        e = eng.Engine(sp, dtype)
        assert not isinstance(e.cast_inputs(xd), eng.WideInputs)          # the binding never takes the uncast route there
        flat = eng.flatten_params(sp, synth.make_params(1, sp.param_shapes()), e.device)
        rows = e._row_buffers(B)
        n = len(rows)
        src = (C.c_void_p * n)(*[xd[t].data_ptr() for t in e.types])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
        pitch = (C.c_int64 * n)(*[r.shape[1] for r in rows])
        out = torch.empty(B * 4, 3, device="cuda")
        ws = e.workspace(B, True)
        rc = e.lib.mshgnn_forward_src(e._plan, 8, src, None, rp, pitch, flat.data_ptr(), out.data_ptr(), ws.data_ptr(), B, 1, None)
        assert rc == -2 and b"wide source rows" in e.lib.mshgnn_last_error()
    e = eng.Engine(spec, "bf16")
    rows = e._row_buffers(B)
    src = (C.c_void_p * 3)(*[xd[t].data_ptr() for t in e.types])
    rp = (C.c_void_p * 3)(*[r.data_ptr() for r in rows])
    pitch = (C.c_int64 * 3)(*[r.shape[1] for r in rows])
    flat = eng.flatten_params(spec, synth.make_params(1, spec.param_shapes()), e.device)
    out = torch.empty(B * 4, 3, device="cuda")
    assert e.lib.mshgnn_forward_src(e._plan, 2, src, None, rp, pitch, flat.data_ptr(), out.data_ptr(), e.workspace(B, True).data_ptr(), B, 1, None) == -1
    assert e.lib.mshgnn_forward_src(e._plan, 8, src, None, None, pitch, flat.data_ptr(), out.data_ptr(), e.workspace(B, True).data_ptr(), B, 1, None) == -1
