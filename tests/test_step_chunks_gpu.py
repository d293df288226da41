"""One-call steps over long batches run as sub-steps over contiguous window ranges (mshgnn_step_mse / mshgnn_step_ce, StepChunk in csrc/mshgnn_device.hpp:
batches of at least twice MSHGNN_STEP_CHUNK, default 32 768, in sub-steps of at least that many windows -- BASELINE configs[3] steps 65 536 windows).  The sub-steps scale their loss terms by the whole batch and the
finalize launches after the first accumulate, so the result is the whole-batch step's up to fp32 summation order: outputs bit-identical (windows are independent),
loss and gradients equal to a few ulps of the accumulated sums.  (The whole step is what the oracle tests pin -- tests/test_full_size_gpu.py runs Solo-12 at
65 536 windows, i.e. chunked, against the oracle on a subsample; a chunked step's stashes are the last sub-step's, so the per-stage helpers do not apply.)"""
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [("c2", "a1-c2", "a1-c2", 3, True, "bf16"), ("c2", "a1-c2", "a1-c2", 3, True, "x3"), ("c2", "a1-c2", "a1-c2", 2, True, "f32"),
         ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 8, False, "bf16"), ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 3, False, "x3"),
         ("k4_com", "solo-k4-com", "solo-k4", 8, True, "bf16"), ("k4_com", "solo-k4-com", "solo-k4", 3, True, "x3"), ("c2", "a1-c2", "a1-c2", 3, False, "bf16")]


def _step(spec, dtype, x_dict, y, params, B):
    from morphsym_hgnn_amd import engine as eng
    e = eng.Engine(spec, dtype)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    n_out = spec.num_nodes[spec.out_type]
    if spec.regression:
        out, loss, g = e.step_mse(xs, flat, y.reshape(-1).to(e.device, torch.float32), B)
    else:
        out, loss, g = e.step_ce(xs, flat, y.reshape(B, n_out).to(e.device, torch.int32).contiguous(), B)
    torch.cuda.synchronize()
    return out.clone(), loss.clone(), g.clone(), e


@pytest.mark.parametrize("kind,topo,cfg,layers,regression,dtype", CASES)
@pytest.mark.parametrize("B,chunk", [(200, 64), (96, 48)])
def test_chunked_step_is_the_whole_step_up_to_summation_order(kind, topo, cfg, layers, regression, dtype, B, chunk, monkeypatch):
    """B = 200 at >= 64 windows per sub-step: three sub-steps of 80 / 80 / 40 windows (equal sub-steps of whole tiles, a ragged last one); B = 96 at 48: two of 48."""
    from morphsym_hgnn_amd import synth
    spec = helpers.make_spec(kind, topo, cfg, 128, layers, regression=regression, grf=3 if kind == "c2" else 1)
    x_dict, y, params = helpers.random_case(spec, B, 11)
    monkeypatch.setenv("MSHGNN_STEP_CHUNK", "0")          # read when the plan is created
    out_w, loss_w, g_w, _ = _step(spec, dtype, x_dict, y, params, B)
    monkeypatch.setenv("MSHGNN_STEP_CHUNK", str(chunk))
    out_c, loss_c, g_c, e = _step(spec, dtype, x_dict, y, params, B)
    assert not e.generic
    assert torch.equal(out_c, out_w)
    assert abs(float(loss_c) - float(loss_w)) <= 2e-6 * abs(float(loss_w))
    from morphsym_hgnn_amd import engine as eng
    ga, gb = eng.unflatten(spec, g_c), eng.unflatten(spec, g_w)
    for k in gb:
        ref = float(gb[k].abs().max())
        if ref == 0.0:
            assert not bool(ga[k].any()), k      # a dead parameter stays an exact zero through the accumulating finalize launches
        else:
            assert float((ga[k] - gb[k]).abs().max()) <= 4e-6 * ref, (k, float((ga[k] - gb[k]).abs().max()) / ref)
    assert not torch.equal(g_c, g_w) or B <= chunk       # (it really ran in sub-steps: the summation order moved some bits)


def test_generic_engine_and_short_batches_run_whole(monkeypatch):
    """The generic-width engine never chunks (its finalize kernel overwrites); neither does a batch under twice the limit: bit-identical to MSHGNN_STEP_CHUNK=0."""
    spec_g = helpers.make_spec("c2", "a1-c2", "a1-c2", 256, 2)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    for sp, B in ((spec_g, 200), (spec, 127)):
        x_dict, y, params = helpers.random_case(sp, B, 3)
        monkeypatch.setenv("MSHGNN_STEP_CHUNK", "0")
        a = _step(sp, "bf16", x_dict, y, params, B)
        monkeypatch.setenv("MSHGNN_STEP_CHUNK", "64")
        b = _step(sp, "bf16", x_dict, y, params, B)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
