"""The specialised one-call step kernels (k_slab_step over the compile-time programs of csrc/mshgnn_spec_tables.inc) against the interpreting kernel of the
same plan: the same MACs in the same order on the same accumulators, so output, loss and every gradient must be IDENTICAL BITS -- for each BASELINE
topology that has a program (A1-C2 at 3 and 8 layers, MiniCheetah-K4 classification, Solo-12 K4 COM), both stash store policies, whole-tile batches;
a ragged batch must fall back to the interpreter.  (Oracle parity of the specialised kernels at the timed sizes: tests/test_full_size_gpu.py, which runs them
by default.)"""
import pytest
import torch

import bench
from morphsym_hgnn_amd import engine as eng, synth

pytestmark = pytest.mark.gpu
CASES = [("a1c2", 3, "A1C2_L3"), ("a1c2", 8, "A1C2_L8"), ("mck4", 8, "MCK4_L8"), ("solo", 8, "SOLO_L8"),
         ("mcc2", 8, "MCC2_L8"), ("solo_s4", 8, "SOLO_S4_L8"), ("mi_quad", 8, "MI_QUAD_L8")]      # (the last three: the model types the reference's classification / COM scripts default to, the MI-HGNN baseline)


def _step(e, spec, x, y, flat, B):
    xs = e.cast_inputs(x)
    if spec.regression:
        out, loss, g = e.step_mse(xs, flat, y.to(e.device, torch.float32).reshape(-1), B)
    else:
        out, loss, g = e.step_ce(xs, flat, y.to(e.device, torch.int32).reshape(B, -1).contiguous(), B)
    torch.cuda.synchronize()
    return out.clone(), loss.clone(), g.clone()


@pytest.mark.parametrize("nt", ["0", "1"])
@pytest.mark.parametrize("config,layers,name", CASES)
def test_specialised_step_is_bit_identical_to_the_interpreter(monkeypatch, config, layers, name, nt):
    spec = bench.build_spec(layers, config)
    monkeypatch.setenv("MSHGNN_SLAB", "2")          # the slab kernels also below one tile per CU (the test batch is small)
    monkeypatch.setenv("MSHGNN_STASH_NT", nt)       # both compile-time store policies
    monkeypatch.setenv("MSHGNN_SPEC", "1")
    e1 = eng.Engine(spec, "bf16")
    monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "bf16")
    assert e1.specialised == name and e0.specialised == "", (e1.specialised, e0.specialised)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e1.device)
    for B in (64, 48, 50):      # whole tiles (unpredicated specialised kernel) ... and a ragged batch (its predicated form)
        x, y = bench.make_batch(spec, B, 11 + B)
        r1 = _step(e1, spec, x, y, flat, B)
        r0 = _step(e0, spec, x, y, flat, B)
        for what, a, b in zip(("out", "loss", "grad"), r1, r0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{name} B={B} nt={nt}: {what} differs, max abs {float((a - b).abs().max())}"
        assert float(r1[2].abs().max()) > 0 and torch.isfinite(r1[1]).all()


X3_CASES = [("a1c2", 3, "X3_A1C2_L3"), ("a1c2", 8, "X3_A1C2_L8"), ("mck4", 8, "X3_MCK4_L8"), ("solo", 8, "X3_SOLO_L8"), ("mcc2", 8, "X3_MCC2_L8"), ("solo_s4", 8, "X3_SOLO_S4_L8"), ("mi_quad", 8, "X3_MI_QUAD_L8")]


@pytest.mark.parametrize("config,layers,name", X3_CASES)
def test_specialised_split_plan_step_is_bit_identical_to_the_interpreter(monkeypatch, config, layers, name):
    """The parity plan's one-call step (k_stack_step_x3) over its compile-time program: predicated stores, so ragged batches run it too."""
    spec = bench.build_spec(layers, config)
    monkeypatch.setenv("MSHGNN_SPEC", "1")
    e1 = eng.Engine(spec, "x3")
    monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "x3")
    assert e1.specialised == name and e0.specialised == "", (e1.specialised, e0.specialised)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e1.device)
    for B in (64, 50, 1040):
        x, y = bench.make_batch(spec, B, 21 + B)
        r1 = _step(e1, spec, x, y, flat, B)
        r0 = _step(e0, spec, x, y, flat, B)
        for what, a, b in zip(("out", "loss", "grad"), r1, r0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{name} B={B}: {what} differs, max abs {float((a - b).abs().max())}"


def test_unlisted_topologies_keep_the_interpreter():
    spec = bench.build_spec(5, "a1c2")      # a depth no program was generated for
    assert eng.Engine(spec, "bf16").specialised == ""


@pytest.mark.parametrize("config,layers", [("a1c2", 3), ("mck4", 8)])
def test_small_batches_take_the_specialised_step_by_default(monkeypatch, config, layers):
    """Below one tile per CU the one-call step used to run the interpreting 8-wave kernel; with a compile-time program it runs the specialised slab kernel at
    every whole-tile batch size (the reference trains at 32 / 64 windows).  Default route == the interpreting slab kernel (MSHGNN_SLAB=2, MSHGNN_SPEC=0) bit for
    bit; against the 8-wave kernels (MSHGNN_SLAB=0) the output is the same bits, loss and gradients agree to fp32 summation order (the loss and the
    decoder's weight gradient are summed over one partial per wave: 4 against 8).  A ragged batch (30 windows: train_classification.py:716) takes the predicated form."""
    spec = bench.build_spec(layers, config)
    monkeypatch.delenv("MSHGNN_SLAB", raising=False); monkeypatch.delenv("MSHGNN_SPEC", raising=False)
    e1 = eng.Engine(spec, "bf16")
    monkeypatch.setenv("MSHGNN_SLAB", "0")
    e8 = eng.Engine(spec, "bf16")
    monkeypatch.setenv("MSHGNN_SLAB", "2"); monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "bf16")
    assert e1.specialised != "" and e0.specialised == ""
    flat = eng.flatten_params(spec, synth.make_params(5, spec.param_shapes()), e1.device)
    for B in (32, 64, 30):
        x, y = bench.make_batch(spec, B, 31 + B)
        r1 = _step(e1, spec, x, y, flat, B)
        r8 = _step(e8, spec, x, y, flat, B)
        r0 = _step(e0, spec, x, y, flat, B)      # (a ragged batch takes the predicated form of the specialised kernel: the same bits again)
        for what, a, b in zip(("out", "loss", "grad"), r1, r0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{config} B={B}: {what} differs from the interpreting slab kernel, max abs {float((a - b).abs().max())}"
        assert torch.equal(r1[0].view(torch.int32), r8[0].view(torch.int32)), f"{config} B={B}: output differs from the 8-wave kernel"
        assert abs(float(r1[1]) - float(r8[1])) <= 1e-6 * abs(float(r8[1])) and float((r1[2] - r8[2]).norm() / r8[2].norm()) < 1e-6


@pytest.mark.parametrize("nt", ["0", "1"])
@pytest.mark.parametrize("config,layers,name", CASES)
def test_specialised_forward_is_bit_identical_to_the_interpreter(monkeypatch, config, layers, name, nt):
    """mshgnn_forward alone (evaluation, and the first call of the two-call training route) over the compile-time programs (k_slab_fwd_spec<.., TR, NT>):
    output bits == the interpreting slab kernel's in evaluation and training mode, and the backward pass that follows a specialised training forward
    (stashes and relu bytes it wrote) gives the same gradient bits.  Whole tiles only: a ragged batch takes the interpreters on both engines."""
    spec = bench.build_spec(layers, config)
    monkeypatch.setenv("MSHGNN_STASH_NT", nt)
    monkeypatch.delenv("MSHGNN_SLAB", raising=False); monkeypatch.setenv("MSHGNN_SPEC", "1")
    e1 = eng.Engine(spec, "bf16")
    monkeypatch.setenv("MSHGNN_SLAB", "2"); monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "bf16")
    assert e1.specialised == name and e0.specialised == ""
    flat = eng.flatten_params(spec, synth.make_params(7, spec.param_shapes()), e1.device)
    for B in (32, 4128, 50):
        x, y = bench.make_batch(spec, B, 41 + B)
        res = []
        for e in (e1, e0):
            xs = e.cast_inputs(x)
            o_eval = e.forward(xs, flat, B, training=False).clone()
            o_tr = e.forward(xs, flat, B, training=True).clone()
            g = e.backward(xs, flat, torch.ones_like(o_tr) / o_tr.numel(), B).clone()
            torch.cuda.synchronize()
            res.append((o_eval, o_tr, g))
        for what, a, b in zip(("eval out", "training out", "grad"), res[0], res[1]):
            if B % 16 == 0:
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{name} B={B} nt={nt}: {what} differs, max abs {float((a - b).abs().max())}"
            else:      # ragged: e1 falls back to the 8-wave kernels below one tile per CU, e0 runs the interpreting slab ones -- same bits up to the decoder's partial sums
                assert float((a - b).norm() / b.norm()) < 1e-6, (name, B, what)
        assert torch.equal(res[0][0].view(torch.int32), res[0][1].view(torch.int32)) and float(res[0][2].abs().max()) > 0


@pytest.mark.parametrize("config,layers,name", X3_CASES)
def test_specialised_split_plan_forward_backward_are_bit_identical_to_the_interpreter(monkeypatch, config, layers, name):
    """The parity plan's forward launch alone (evaluation; first call of the two-call route -- what the nn.Module surface, default precision "x3", runs) and its backward launch
    alone over the compile-time programs (k_stack_fwd_x3_spec / k_stack_bwd_x3_spec): predicated stores, so ragged batches too.  Bits == the interpreting kernels'."""
    spec = bench.build_spec(layers, config)
    monkeypatch.setenv("MSHGNN_SPEC", "1")
    e1 = eng.Engine(spec, "x3")
    monkeypatch.setenv("MSHGNN_SPEC", "0")
    e0 = eng.Engine(spec, "x3")
    assert e1.specialised == name and e0.specialised == ""
    flat = eng.flatten_params(spec, synth.make_params(9, spec.param_shapes()), e1.device)
    for B in (32, 50, 1040):
        x, y = bench.make_batch(spec, B, 51 + B)
        res = []
        for e in (e1, e0):
            xs = e.cast_inputs(x)
            o_eval = e.forward(xs, flat, B, training=False).clone()
            o_tr = e.forward(xs, flat, B, training=True).clone()
            g = e.backward(xs, flat, torch.ones_like(o_tr) / o_tr.numel(), B).clone()
            torch.cuda.synchronize()
            res.append((o_eval, o_tr, g))
        for what, a, b in zip(("eval out", "training out", "grad"), res[0], res[1]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{name} B={B}: {what} differs, max abs {float((a - b).abs().max())}"
        assert float(res[0][2].abs().max()) > 0
