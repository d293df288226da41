"""On-device window assembly: oracle vs the committed fixture (values produced by the reference's
own dataset functions, oracle/gen_window_golden.py), and the HIP gather kernel through the C-ABI vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import window_oracle as wo
from oracle.gen_window_golden import synthetic_sequence, CASES

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "windows_a1c2.npz"))
T, N = int(FX["T"]), int(FX["N"])
SEQ = synthetic_sequence(int(FX["seed"]), N)
JP, FP = FX["joint_perm"].astype(int), FX["foot_perm"].astype(int)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_fixture(case):
    for st in FX["starts"]:
        b, j, f, y, q = wo.a1_c2_window(SEQ, int(st), T, JP, FP, case["grf"], case["body"], case["norm"])
        tol = 1e-12 if (case["body"] or case["norm"]) else 0.0
        key = f"{case['name']}:{int(st)}"
        assert np.abs(y - FX[key + ":y"]).max() <= tol * max(1.0, np.abs(y).max())
        assert np.abs(b[:, ::7] - FX[key + ":base"]).max() <= tol and np.abs(j[:, ::11] - FX[key + ":joint"]).max() <= tol
        assert b.shape == (2, 6 * T) and j.shape == (12, 3 * T) and f.shape == (4, 1) and q.shape == (4,)


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [True, False], ids=["chunk-gather", "run-gather"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_device_assembly_matches_oracle(case, dtype, fast):
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe
    if case["grf"] == 1 and case["body"]:
        pytest.skip("not a reference configuration")
    recipe = quadsdk_a1_c2_recipe(JP, FP, T, case["grf"], case["body"], case["norm"])
    store = SequenceStore(SEQ, recipe, dtype=dtype, fast=fast)      # (both gather kernels; standardised recipes always take the run-by-run one)
    assert len(store) == N - T + 1
    starts = [0, 1, 37, 249, 250, 17, 17, 250, 3]            # repeats and the last valid window
    xs, y, q = store.assemble(starts)
    B = len(starts)
    want = [wo.a1_c2_window(SEQ, s, T, JP, FP, case["grf"], case["body"], case["norm"]) for s in starts]
    for ti, (name, n) in enumerate((("base", 2), ("joint", 12), ("foot", 4))):
        ref = np.stack([w[ti] for w in want]).reshape(B * n, -1)
        got = xs[ti].float().cpu().numpy().astype(np.float64)
        F = ref.shape[1]
        assert got.shape[0] == B * n and got.shape[1] >= F and not got[:, F:].any()          # pad columns are zero
        if dtype == "f32":
            tol = 0.0 if not case["norm"] else 2e-7 * np.abs(ref).max()      # raw values are fp32-exact; standardised ones round once
            assert np.abs(got[:, :F] - ref).max() <= tol, name
        else:
            want_bf = torch.from_numpy(ref).to(torch.bfloat16).double().numpy()
            # bit-exact bf16 rounding of the fp32 value (standardised: fp64 -> fp32 -> bf16, may differ from fp64 -> bf16 by 1 ulp)
            assert np.abs(got[:, :F] - want_bf).max() <= (0.0 if not case["norm"] else 2.0 ** -7 * np.abs(ref).max()), name
    ref_y = np.stack([w[3] for w in want])
    assert np.abs(y.cpu().numpy() - ref_y).max() <= (1e-6 * np.abs(ref_y).max() if case["body"] else 0.0)
    assert np.abs(q.cpu().numpy() - np.stack([w[4] for w in want])).max() == 0.0
    with pytest.raises(IndexError):
        store.assemble([N - T + 1])


@pytest.mark.gpu
def test_assembled_windows_feed_the_engine():
    """Windows gathered on device go straight into Engine.forward and give the same output as the reference-convention
    x_dict built by the oracle and cast by Engine.cast_inputs."""
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe
    from tests import helpers
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    e = eng.Engine(spec, "f32")
    store = SequenceStore(SEQ, quadsdk_a1_c2_recipe(JP, FP, T, 3), dtype="f32")
    starts = list(range(0, 240, 7))
    B = len(starts)
    xs, y, _ = store.assemble(starts)
    flat = eng.flatten_params(spec, synth.make_params(4, spec.param_shapes()), e.device)
    out_dev = e.forward(xs, flat, B, training=False).clone()
    want = [wo.a1_c2_window(SEQ, s, T, JP, FP, 3) for s in starts]
    x_dict = {t: torch.from_numpy(np.stack([w[i] for w in want]).reshape(B * n, -1)) for i, (t, n) in enumerate((("base", 2), ("joint", 12), ("foot", 4)))}
    out_ref = e.forward(e.cast_inputs(x_dict), flat, B, training=False)
    assert torch.equal(out_dev, out_ref)


# --- MiniCheetah K4 (contact classification data format) ----------------------------------------
from oracle.gen_window_golden import minicheetah_sequence   # noqa: E402

FX4 = np.load(os.path.join(os.path.dirname(__file__), "golden", "windows_mck4.npz"))
SEQ4 = minicheetah_sequence(int(FX4["seed"]), int(FX4["N"]))


def test_k4_oracle_matches_reference_fixture():
    for st in FX4["starts"]:
        b, j, f, y = wo.minicheetah_k4_window(SEQ4, int(st), T, JP, FP)
        k = f"k4:{int(st)}"
        assert np.array_equal(y, FX4[k + ":y"]) and np.array_equal(b[:, ::7], FX4[k + ":base"])
        assert np.array_equal(j[:, ::11], FX4[k + ":joint"]) and np.array_equal(f[:, ::13], FX4[k + ":foot"])
        assert b.shape == (4, 6 * T) and j.shape == (12, 2 * T) and f.shape == (4, 6 * T)


def test_k4_standardised_windows_match_the_reference_branch():
    """normalize=True (what BASELINE configs[2] trains with): the oracle against the vectors the REFERENCE'S OWN standardisation branch produced
    (LinTzuYaunDataset_Morph.py:337-345, run by oracle/gen_window_golden.py under a numpy-1.x np.nan_to_num shim; the A1 counterpart is the d3_norm case above)."""
    for st in FX4["starts"]:
        b, j, f, y = wo.minicheetah_k4_window(SEQ4, int(st), T, JP, FP, normalize=True)
        k = f"k4_norm:{int(st)}"
        for got, key in ((y, ":y"), (b[:, ::7], ":base"), (j[:, ::11], ":joint"), (f[:, ::13], ":foot")):
            ref = FX4[k + key]
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0), (k, key)


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [False, True])
def test_k4_device_assembly_matches_oracle(normalize):
    from morphsym_hgnn_amd.windows import SequenceStore, minicheetah_k4_recipe
    store = SequenceStore(SEQ4, minicheetah_k4_recipe(JP, FP, T, normalize), dtype="f32")
    starts = [250, 0, 99, 99, 13]
    xs, y, q = store.assemble(starts)
    want = [wo.minicheetah_k4_window(SEQ4, s, T, JP, FP, normalize) for s in starts]
    for ti, n in enumerate((4, 12, 4)):
        ref = np.stack([w[ti] for w in want]).reshape(len(starts) * n, -1)
        got = xs[ti].cpu().numpy().astype(np.float64)[:, :ref.shape[1]]
        assert np.abs(got - ref).max() <= (0.0 if not normalize else 2e-7 * np.abs(ref).max())
    assert np.array_equal(y.cpu().numpy(), np.stack([w[3] for w in want])) and q is None



# --- Solo-12 centroidal-momentum task (SURVEY.md 8(d) config 4 data format; soloDataset.py) ------------------------------------------
from oracle.gen_window_golden import solo_sequence, SOLO_KINDS   # noqa: E402

FXS = np.load(os.path.join(os.path.dirname(__file__), "golden", "windows_solo.npz"))
SEQS = solo_sequence(int(FXS["seed"]), int(FXS["N"]))


@pytest.mark.parametrize("kind", list(SOLO_KINDS))
def test_solo_oracle_matches_reference_fixture(kind):
    nb = SOLO_KINDS[kind][1]
    for Th in (1, 5):
        for st in FXS["starts"]:
            b, j, y = wo.solo_com_window(SEQS["X"], SEQS["Y"], int(st), Th, JP, nb)
            k = f"{kind}:{Th}:{int(st)}"
            assert np.array_equal(y, FXS[k + ":y"]) and np.array_equal(j, FXS[k + ":joint"])
            assert b.shape == (nb, 6 * Th) and not b.any() and j.shape == (12, 2 * Th) and y.shape == (6 * nb,)


@pytest.mark.gpu
@pytest.mark.parametrize("history", [1, 5])
@pytest.mark.parametrize("kind", list(SOLO_KINDS))
def test_solo_device_assembly_matches_oracle_and_feeds_the_engine(kind, history):
    from morphsym_hgnn_amd.windows import SequenceStore, solo_com_recipe, solo_com_arrays
    nb = SOLO_KINDS[kind][1]
    store = SequenceStore(solo_com_arrays(SEQS["X"], SEQS["Y"]), solo_com_recipe(kind, JP, history), dtype="f32")
    N4 = int(FXS["N"])
    starts = [0, N4 - history, 17, 17, 3]
    xs, y, q = store.assemble(starts)
    want = [wo.solo_com_window(SEQS["X"], SEQS["Y"], s, history, JP, nb) for s in starts]
    for ti, n in enumerate((nb, 12)):
        ref = np.stack([w[ti] for w in want]).reshape(len(starts) * n, -1)
        got = xs[ti].cpu().numpy().astype(np.float64)
        assert np.array_equal(got[:, :ref.shape[1]], ref) and not got[:, ref.shape[1]:].any()
    assert np.array_equal(y.cpu().numpy(), np.stack([w[2] for w in want])) and q is None
    if history == 1 and kind != "s4_com":      # the COM models' setting: the assembled batch goes straight into the engine
        from morphsym_hgnn_amd import engine as eng, synth
        from tests import helpers
        topo, cfg = {"k4_com": ("solo-k4-com", "solo-k4"), "c2_com": ("solo-c2-com", "solo-c2")}[kind]
        spec = helpers.make_spec(kind, topo, cfg, 128, 2)
        e = eng.Engine(spec, "f32")
        flat = eng.flatten_params(spec, synth.make_params(4, spec.param_shapes()), e.device)
        B = len(starts)
        out_dev = e.forward(xs, flat, B, training=False).clone()
        x_dict = {t: torch.from_numpy(np.stack([w[i] for w in want]).reshape(B * n, -1)) for i, (t, n) in enumerate((("base", nb), ("joint", 12)))}
        assert torch.equal(out_dev, e.forward(e.cast_inputs(x_dict), flat, B, training=False))


# --- fused window assembly: the encoder gathers its inputs from the resident series (mshgnn_step_mse_series) -------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("recipe_name,B", [("a1c2", 37), ("a1c2", 1000), ("a1c2_body", 64), ("mck4", 130), ("mck4", 1)])
def test_split_plan_series_step_is_bit_identical_to_assemble_then_step(recipe_name, B):
    """The same on the split-bf16 parity plan: its encoder gathers fp32 inputs from the fp32 series (two 16-byte loads per chunk, a register
    splice where a chunk straddles two runs) and materialises fp32 windows; regression (fused MSE) and contact classification."""
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe, minicheetah_k4_recipe
    from tests import helpers
    if recipe_name.startswith("a1c2"):
        seq, n = SEQ, N
        recipe = quadsdk_a1_c2_recipe(JP, FP, T, 3, body_frame_labels=recipe_name.endswith("body"))
        spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    else:
        seq, n = dict(SEQ4), int(FX4["N"])
        recipe = minicheetah_k4_recipe(JP, FP, T)
        spec = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 3, regression=False)
    store = SequenceStore(seq, recipe, dtype="x3")
    e = eng.Engine(spec, "x3")
    assert not e.generic
    starts = torch.randint(0, n - T + 1, (B,), generator=torch.Generator().manual_seed(B))
    starts[0], starts[-1] = 0, n - T
    starts = starts.cuda()
    flat = eng.flatten_params(spec, synth.make_params(8, spec.param_shapes()), e.device)
    xs, y, _ = store.assemble(starts)
    xs = [x.clone() for x in xs]; y = y.clone()
    if spec.regression:
        out_a, loss_a, g_a = e.step_mse(xs, flat, y.reshape(-1), B)
    else:
        out_a, loss_a, g_a = e.step_ce(xs, flat, (y != 0).to(torch.int32).reshape(B, 4).contiguous(), B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    for x, t in zip(store._buffers(B)[0], recipe.node_types):
        x.fill_(float("nan"))                    # the fused step must rewrite every feature column ...
        x[:, recipe.width(t):] = 0               # ... and leaves the (zero) pad columns alone
    xs2, y2, out_b, loss_b, g_b = (e.step_mse_series if spec.regression else e.step_ce_series)(store, starts, flat)
    torch.cuda.synchronize()
    for a, b in zip(xs, xs2):
        assert torch.equal(a, b)
    assert torch.equal(out_a, out_b) and torch.equal(loss_a, loss_b) and torch.equal(g_a, g_b)
    assert torch.equal(y, store._buffers(B)[1])
    with pytest.raises(ValueError, match="materialised"):
        e.step_mse_series(store, starts, flat, materialize=False)


@pytest.mark.gpu
@pytest.mark.parametrize("recipe_name,B", [("a1c2", 37), ("a1c2", 1000), ("a1c2_body", 64), ("mck4", 130)])
def test_series_step_is_bit_identical_to_assemble_then_step(recipe_name, B):
    """One training step straight from the sequence (window gather fused into the encoder) == mshgnn_assemble_windows + mshgnn_step_mse: the
    materialised windows, labels, outputs, loss and every gradient are the same bits (random starts incl. the first and the last window)."""
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe, minicheetah_k4_recipe
    from tests import helpers
    if recipe_name.startswith("a1c2"):
        seq, n = SEQ, N
        recipe = quadsdk_a1_c2_recipe(JP, FP, T, 3, body_frame_labels=recipe_name.endswith("body"))
        spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    else:
        seq, n = dict(SEQ4), int(FX4["N"])
        seq["F"] = SEQ["F"][:n, :4]                         # a regression label series for the K4 graph (4 feet x 1)
        recipe = minicheetah_k4_recipe(JP, FP, T)
        recipe.label_series, recipe.label_cols = "F", [0, 1, 2, 3]
        spec = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 3, grf=1)
    store = SequenceStore(seq, recipe, dtype="bf16")
    e = eng.Engine(spec, "bf16")
    g = torch.Generator().manual_seed(B)
    starts = torch.randint(0, n - T + 1, (B,), generator=g)
    starts[0], starts[-1] = 0, n - T
    starts = starts.cuda()
    flat = eng.flatten_params(spec, synth.make_params(8, spec.param_shapes()), e.device)
    xs, y, _ = store.assemble(starts)
    xs = [x.clone() for x in xs]; y = y.clone()
    out_a, loss_a, g_a = e.step_mse(xs, flat, y.reshape(-1), B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    xs2, y2, out_b, loss_b, g_b = e.step_mse_series(store, starts, flat, materialize=True)
    torch.cuda.synchronize()
    assert store.desc.run_ptrs_ready == 0                   # the first step of a store resolves the runs' column pointers ...
    for a, b in zip(xs, xs2):
        assert torch.equal(a, b)
    assert torch.equal(y, y2) and torch.equal(out_a, out_b) and torch.equal(loss_a, loss_b) and torch.equal(g_a, g_b)
    # ... and without materialised windows at all: the weight-gradient kernel gathers its raw operands from the series too
    out_b, loss_b, g_b = out_b.clone(), loss_b.clone(), g_b.clone()
    xs3, y3, out_c, loss_c, g_c = e.step_mse_series(store, starts, flat, materialize=False)
    torch.cuda.synchronize()
    assert store.desc.run_ptrs_ready == 1                   # ... later steps on the same stream vouch for the scratch (mshgnn_window_desc.run_ptrs_ready): same bits
    assert xs3 is None and torch.equal(y, y3) and torch.equal(out_a, out_c) and torch.equal(loss_a, loss_c) and torch.equal(g_a, g_c)


@pytest.mark.gpu
@pytest.mark.parametrize("materialize", [True, False])
def test_series_step_takes_node_rows_of_more_than_16_runs(materialize):
    """A node row built from more than 16 T-long variables (18 here: the fused-gather encoder keeps the first 16 run pointers in LDS and reads the
    rest from the global table) runs through mshgnn_step_mse_series with the same bits as assembly + mshgnn_step_mse (the route
    used to refuse such recipes)."""
    from morphsym_hgnn_amd import engine as eng, synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe
    hist, B = 10, 77
    recipe = quadsdk_a1_c2_recipe(JP, FP, hist, 3, n_base=1)
    recipe.variables["base"] = recipe.variables["base"] * 3            # 18 runs of 10 steps per base row
    assert recipe.width("base") == 180 and recipe.width("base") // hist > 16
    spec = ModelSpec(kind="mi", topology=topology.TOPOLOGIES["quadruped-mi"](), hidden=128, num_layers=2,
                     widths={t: recipe.width(t) for t in recipe.node_types}, regression=True, grf_dimension=3, group=None, num_timesteps=hist)
    store = SequenceStore(SEQ, recipe, dtype="bf16")
    e = eng.Engine(spec, "bf16")
    g = torch.Generator().manual_seed(5)
    starts = torch.randint(0, N - hist + 1, (B,), generator=g)
    starts[0], starts[-1] = 0, N - hist
    starts = starts.cuda()
    flat = eng.flatten_params(spec, synth.make_params(8, spec.param_shapes()), e.device)
    xs, y, _ = store.assemble(starts)
    xs = [x.clone() for x in xs]; y = y.clone()
    out_a, loss_a, g_a = e.step_mse(xs, flat, y.reshape(-1), B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    xs2, y2, out_b, loss_b, g_b = e.step_mse_series(store, starts, flat, materialize=materialize)
    torch.cuda.synchronize()
    if materialize:
        for a, b in zip(xs, xs2):
            assert torch.equal(a, b)
    assert torch.equal(y, y2) and torch.equal(out_a, out_b) and torch.equal(loss_a, loss_b) and torch.equal(g_a, g_b)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 130, 1000])
def test_classification_series_step_is_bit_identical_to_assemble_then_step_ce(B):
    """MiniCheetah-K4 contact classification straight from the sequence (mshgnn_step_ce_series): labels = the contact flags of each window's
    last step; same bits as mshgnn_assemble_windows + mshgnn_step_ce, with and without materialised windows."""
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, minicheetah_k4_recipe
    from tests import helpers
    seq, n = dict(SEQ4), int(FX4["N"])
    recipe = minicheetah_k4_recipe(JP, FP, T)
    spec = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 3, regression=False)
    store = SequenceStore(seq, recipe, dtype="bf16")
    e = eng.Engine(spec, "bf16")
    starts = torch.randint(0, n - T + 1, (B,), generator=torch.Generator().manual_seed(B))
    starts[0], starts[-1] = 0, n - T
    starts = starts.cuda()
    flat = eng.flatten_params(spec, synth.make_params(8, spec.param_shapes()), e.device)
    xs, y, _ = store.assemble(starts)
    xs = [x.clone() for x in xs]
    lab = (y != 0).to(torch.int32).reshape(B, 4).contiguous()
    assert 0 < int(lab.sum()) < lab.numel()
    out_a, loss_a, g_a = e.step_ce(xs, flat, lab, B)
    out_a, loss_a, g_a = out_a.clone(), loss_a.clone(), g_a.clone()
    for mat in (True, False):
        xs2, lab2, out_b, loss_b, g_b = e.step_ce_series(store, starts, flat, materialize=mat)
        torch.cuda.synchronize()
        if mat:
            assert all(torch.equal(a, b) for a, b in zip(xs, xs2))
        assert torch.equal(lab, lab2) and torch.equal(out_a, out_b) and torch.equal(loss_a, loss_b) and torch.equal(g_a, g_b)
    # a regression plan refuses the classification entry point and the other way round
    with pytest.raises(eng.MshgnnError):
        e.step_mse_series(store, starts, flat)


@pytest.mark.gpu
def test_series_step_refuses_what_it_cannot_run():
    """mshgnn_step_mse_series is a route of the bf16 and the split plan: the fp32 plan, standardised recipes and recipes whose node types differ from the plan's are
    refused with an error (the caller assembles windows and calls mshgnn_step_mse instead) -- never a silent fallback."""
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe, minicheetah_k4_recipe
    from tests import helpers
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    starts = torch.tensor([0, 5, 9], dtype=torch.int64).cuda()
    flat32 = None
    for dtype, recipe, match in (("f32", quadsdk_a1_c2_recipe(JP, FP, T, 3), "bf16 plan"),
                                 ("bf16", quadsdk_a1_c2_recipe(JP, FP, T, 3, normalize=True), "unstandardised"),
                                 ("bf16", quadsdk_a1_c2_recipe(JP, FP, T, 1), "label count")):
        e = eng.Engine(spec, dtype)
        store = SequenceStore(SEQ, recipe, dtype="bf16" if dtype == "bf16" else "f32")
        flat = eng.flatten_params(spec, synth.make_params(1, spec.param_shapes()), e.device)
        with pytest.raises(eng.MshgnnError, match=match):
            e.step_mse_series(store, starts, flat)
