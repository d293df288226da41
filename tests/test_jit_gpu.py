"""Programs compiled on demand (morphsym_hgnn_amd/jit.py + mshgnn_plan_attach_program): a topology / depth the build has no compile-time program for gets the library's own
kernel source compiled over ITS plan tables (hipcc, shard 99) and attached to the plan.  Same MACs in the same order as the interpreting kernels: identical bits on every
route (one-call step, evaluation forward, two-call training), whole-tile and ragged batches; a program of other tables is refused and changes nothing.
(__graft_entry__.build() warms the cache for the plan used here; without that the first test compiles for about a minute.)"""
import pytest
import torch

import bench
from morphsym_hgnn_amd import engine as eng, jit, synth

pytestmark = pytest.mark.gpu


def _routes(e, spec, x, y, flat, B):
    xs = e.cast_inputs(x)
    out, loss, g = e.step_mse(xs, flat, y.to(e.device, torch.float32).reshape(-1), B)
    res = [out.clone(), loss.clone(), g.clone()]
    res.append(e.forward(xs, flat, B, training=False).clone())
    o_tr = e.forward(xs, flat, B, training=True).clone()
    res.append(e.backward(xs, flat, torch.ones_like(o_tr) / o_tr.numel(), B).clone())
    torch.cuda.synchronize()
    return res


def test_program_compiled_on_demand_is_bit_identical_to_the_interpreter(monkeypatch):
    monkeypatch.delenv("MSHGNN_JIT", raising=False); monkeypatch.delenv("MSHGNN_SPEC", raising=False)
    spec = bench.build_spec(5, "a1c2")      # a depth the build has no program for
    monkeypatch.setenv("MSHGNN_SLAB", "2")
    e0 = eng.Engine(spec, "bf16")           # the interpreting slab kernels
    monkeypatch.delenv("MSHGNN_SLAB")
    e1 = eng.Engine(spec, "bf16")
    assert e0.specialised == "" and e1.specialised == ""
    name = jit.attach_program(e1)
    assert name.startswith("JIT_") and e1.specialised == name
    assert jit.attach_program(e1) == name      # (a plan that has a program keeps it)
    flat = eng.flatten_params(spec, synth.make_params(11, spec.param_shapes()), e1.device)
    for B in (64, 50, 4128):
        x, y = bench.make_batch(spec, B, 71 + B)
        r1, r0 = _routes(e1, spec, x, y, flat, B), _routes(e0, spec, x, y, flat, B)
        for what, a, b in zip(("step out", "step loss", "step grad", "eval out", "two-call grad"), r1, r0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"B={B}: {what} differs, max abs {float((a - b).abs().max())}"
        assert float(r1[2].abs().max()) > 0


def test_program_of_other_tables_is_refused():
    spec5, spec4 = bench.build_spec(5, "a1c2"), bench.build_spec(4, "a1c2")
    e5, e4 = eng.Engine(spec5, "bf16"), eng.Engine(spec4, "bf16")
    jit.attach_program(e5)
    selector = eng.C.cast(e5._jit_lib.mshgnn_jit_program, eng.C.c_void_p)
    assert e4.lib.mshgnn_plan_attach_program(e4._plan, selector) != 0
    assert b"not this plan" in e4.lib.mshgnn_last_error()
    assert e4.lib.mshgnn_plan_specialised(e4._plan) == b""      # nothing changed: the plan still interprets
    x, y = bench.make_batch(spec4, 32, 3)
    flat = eng.flatten_params(spec4, synth.make_params(1, spec4.param_shapes()), e4.device)
    out, loss, g = e4.step_mse(e4.cast_inputs(x), flat, y.to(e4.device, torch.float32).reshape(-1), 32)
    assert torch.isfinite(loss).all()
    with pytest.raises(RuntimeError):      # the exact-fp32 plan (per-layer kernels) takes no program
        jit.attach_program(eng.Engine(spec4, "f32"))


def test_split_plan_program_compiled_on_demand_is_bit_identical_to_the_interpreter(monkeypatch):
    monkeypatch.delenv("MSHGNN_JIT", raising=False)
    spec = bench.build_spec(5, "a1c2")
    monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "x3")
    monkeypatch.delenv("MSHGNN_SPEC")
    e1 = eng.Engine(spec, "x3")
    assert e1.specialised == ""
    assert jit.attach_program(e1).startswith("JIT_X3_")
    flat = eng.flatten_params(spec, synth.make_params(13, spec.param_shapes()), e1.device)
    for B in (32, 50):
        x, y = bench.make_batch(spec, B, 91 + B)
        r1, r0 = _routes(e1, spec, x, y, flat, B), _routes(e0, spec, x, y, flat, B)
        for what, a, b in zip(("step out", "step loss", "step grad", "eval out", "two-call grad"), r1, r0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"B={B}: {what} differs, max abs {float((a - b).abs().max())}"


def test_environment_switch_attaches_at_plan_creation(monkeypatch):
    monkeypatch.setenv("MSHGNN_JIT", "1")
    e = eng.Engine(bench.build_spec(5, "a1c2"), "bf16")
    assert e.specialised.startswith("JIT_")
    assert eng.Engine(bench.build_spec(3, "a1c2"), "bf16").specialised == "A1C2_L3"      # built-in programs are not replaced
