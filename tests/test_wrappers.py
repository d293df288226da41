"""The training-step wrappers (morphsym_hgnn_amd/wrappers.py) against the golden vectors: `training_step` returns the loss the reference's
Lightning modules return (gnnLightning.py:179-186, 709-722) and `loss.backward()` leaves the golden parameter gradients -- with the loss
and its gradient coming from the device metric kernels, not from torch's loss arithmetic."""
import types

import pytest
import torch

from tests import helpers


def _batch(x_dict, ei, y, B, dev, r_o=None):
    b = types.SimpleNamespace(x_dict={k: v.to(dev) for k, v in x_dict.items()}, edge_index_dict={k: v.to(dev) for k, v in ei.items()},
                              y=y.to(dev).flatten(), batch_size=B)
    if r_o is not None:
        b.r_o = r_o.to(dev).flatten()
    return b


def _wrapper(case, spec, dummy, **kw):
    from morphsym_hgnn_amd import wrappers
    _, cfg = helpers.load_group(case["cfg"])
    meta = spec.topology.metadata()
    sym = dict(symmetry_mode="MorphSym" if cfg else None, group_operator_path=cfg)
    if case["kind"] == "c2" and case["regression"]:
        return wrappers.HGNN_C2_Lightning_Reg(case["hidden"], case["layers"], meta, dummy, lr=1e-4, grf_dimension=case["grf"], **sym, **kw)
    if case["kind"] == "c2":
        return wrappers.HGNN_C2_Lightning_Cls(case["hidden"], case["layers"], meta, dummy, lr=1e-4, regression=False, **sym, **kw)
    if case["kind"] == "k4":
        return wrappers.HGNN_K4_Lightning(case["hidden"], case["layers"], meta, dummy, lr=1e-4, regression=case["regression"], **sym, **kw)
    return wrappers.Heterogeneous_GNN_Lightning(case["hidden"], case["layers"], meta, dummy, lr=1e-4, regression=case["regression"],
                                                grf_dimension=case["grf"], **kw)


def test_optimizer_setting_and_missing_helper_raise_like_the_reference():
    from morphsym_hgnn_amd import wrappers
    w = wrappers.Base_Lightning("lbfgs", 1e-3, True)
    w.lin = torch.nn.Linear(2, 2)
    with pytest.raises(ValueError, match="Invalid optimizer setting"):       # gnnLightning.py:264
        w.configure_optimizers()
    w.optimizer = "sgd"
    assert isinstance(w.configure_optimizers(), torch.optim.SGD)
    w.optimizer = "adam"
    opt = w.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-3
    with pytest.raises(NotImplementedError):
        w.step_helper_function(None)
    assert w.mse_loss is None and w.ce_loss is None          # before the first step, as in the reference's constructor (:65-74)
    p = torch.tensor([[0.2, -1.0, 3.0, 0.5, 0.1, 0.0, -2.0, 2.0]])
    per_foot, prob, only1 = w.classification_calculate_useful_values(p, 1)
    assert per_foot.shape == (4, 2) and torch.allclose(prob.sum(1), torch.ones(4)) and only1.shape == (1, 4)


@pytest.fixture(params=["f32", "x3"])
def parity_plan(request, monkeypatch):
    """The two parity-grade plans (north_star tolerance 1e-4): exact fp32 MFMA and split-bf16; the wrappers' models read MSHGNN_DTYPE."""
    monkeypatch.setenv("MSHGNN_DTYPE", request.param)
    return request.param


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["a1c2_h128_L3_d3_B3", "mck4_cls_h128_L2_B3", "mi_h128_L2_d1_B3", "mcc2_cls_h128_L2_B3"])
def test_training_step_returns_the_golden_loss_and_backward_the_golden_gradients(name, fused, parity_plan):
    """Both routes of training_step: the one-call engine step (labels handed to the model: forward + loss + backward in one C-ABI call, the
    returned loss delivers the gradients) and forward -> device metric loss -> autograd."""
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    dev = torch.device("cuda")
    batch = _batch(x_dict, ei, y, B, dev)
    w = _wrapper(case, spec, batch).to(dev)
    w.model.load_state_dict(params)
    w.fused_training_step = fused
    loss = w.training_step(batch, 0)
    assert w.model._gpend_id == (1 if fused else 0)            # the route that was asked for is the one that ran
    assert next(iter(w.model._engines.values())).storage == parity_plan
    assert loss.requires_grad and loss is (w.mse_loss if case["regression"] else w.ce_loss)
    w.model.zero_grad()                                        # (Lightning clears the gradients between training_step and backward)
    loss.backward()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in w.model.named_parameters()}
    out = w.model(x_dict=batch.x_dict, edge_index_dict=batch.edge_index_dict)
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().double().cpu(), grads, rtol=1e-4, what=name)
    # accumulation and an upstream factor: a second step without clearing adds 0.5 x the same gradient
    g1 = w.model.decoder.weight.grad.detach().clone()
    (0.5 * w.training_step(batch, 1)).backward()
    assert float((w.model.decoder.weight.grad - 1.5 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    w.model.zero_grad()
    w.training_step(batch, 2).backward()
    key = "train_MSE_loss" if case["regression"] else "train_CE_loss"
    assert torch.equal(w.logged[key], loss)
    if not case["regression"]:
        assert 0.0 <= float(w.logged["train_Accuracy"]) <= 1.0 and "train_F1_Score_Leg_3" in w.logged
    # validation: epoch values are the ratio of the accumulated sums; one optimizer step runs on the view parameters
    w.on_validation_epoch_start()
    with torch.no_grad():        # (as a Trainer runs validation)
        v1 = w.validation_step(batch, 0); v2 = w.validation_step(batch, 1)
    assert not v1.requires_grad
    w.on_validation_epoch_end()
    ep = w.logged["val_MSE_loss" if case["regression"] else "val_CE_loss"]
    assert abs(float(ep) - 0.5 * (float(v1) + float(v2))) <= 1e-6 * abs(float(ep))
    opt = w.configure_optimizers()
    before = w.model.decoder.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, w.model.decoder.weight.detach())


@pytest.mark.gpu
def test_world_frame_metrics_follow_the_rotated_forces():
    """HGNN_C2_Lightning_Reg(grf_body_to_world_frame=True): the returned loss stays the body-frame MSE, the *_WorldFrame values are the
    metrics of the rotated labels / predictions (gnnLightning.py:621-631, 663-676) -- rotation invariance makes MSE equal in both frames."""
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    B = case["B"]
    dev = torch.device("cuda")
    q = torch.randn(B, 4, generator=torch.Generator().manual_seed(0))
    batch = _batch(x_dict, ei, y, B, dev, r_o=q)
    w = _wrapper(case, spec, batch, grf_body_to_world_frame=True).to(dev)
    w.model.load_state_dict(params)
    loss = w.training_step(batch, 0)
    assert loss.requires_grad
    loss = loss.detach()
    assert abs(float(w.mse_loss_worldframe) - float(loss)) <= 1e-5 * float(loss)          # rotations preserve the squared error
    assert abs(float(w.logged["train_RMSE_loss_WorldFrame"]) - float(loss) ** 0.5) <= 1e-5 * float(loss) ** 0.5
    assert float(w.l1_loss_worldframe) > 0 and abs(float(w.l1_loss_worldframe) - float(w.l1_loss)) > 0      # L1 is frame dependent


@pytest.mark.gpu
@pytest.mark.parametrize("name,model_type", [("solok4com_h128_L3_B5", "heterogeneous_gnn_k4_com"), ("soloc2com_h128_L2_B4", "heterogeneous_gnn_c2_com"),
                                             ("solos4com_h128_L2_B3", "heterogeneous_gnn_s4_com"), ("com_hgnn_h128_L2_B3", None)])
@pytest.mark.parametrize("fused", [True, False])
def test_com_training_step_returns_the_golden_loss_and_gradients(name, model_type, fused):
    """COM_HGNN_SYM_Lightning / COM_HGNN_Lightning (gnnLightning_com.py:290-409): loss = MSE over [batch, num_bases * 6], golden gradients;
    the extra metrics (lin / ang MSE, cosine similarities) against the oracle."""
    from morphsym_hgnn_amd import wrappers
    from oracle import metrics_oracle as mo
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    B = case["B"]
    dev = torch.device("cuda")
    batch = _batch(x_dict, ei, y, B, dev)
    _, cfg = helpers.load_group(case["cfg"])
    stats = ([0.1, -0.2, 0.3, 0.0, 0.5, -0.1], [1.5, 0.7, 1.1, 2.0, 0.9, 1.3])
    meta = spec.topology.metadata()
    if model_type is None:
        w = wrappers.COM_HGNN_Lightning(case["hidden"], case["layers"], meta, batch, lr=1e-4, stats=stats)
    else:
        w = wrappers.COM_HGNN_SYM_Lightning(case["hidden"], case["layers"], meta, batch, lr=1e-4, symmetry_mode="MorphSym" if cfg else None,
                                            group_operator_path=cfg, model_type=model_type, stats=stats)
    w = w.to(dev)
    w.model.load_state_dict(params)
    w.fused_training_step = fused
    loss = w.training_step(batch, 0)
    assert w.model._gpend_id == (1 if fused else 0)
    assert loss.requires_grad and loss is w.loss
    loss.backward()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in w.model.named_parameters()}
    out = w.model(x_dict=batch.x_dict, edge_index_dict=batch.edge_index_dict)
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().double().cpu(), grads, rtol=1e-4, what=name)
    nb = w.model.num_bases
    o = mo.com_metrics(y.reshape(B, -1).float().numpy(), out.detach().reshape(B, -1).float().cpu().numpy(), nb, *stats)
    for key, okey in (("train_MSE_loss_lin", "mse_lin"), ("train_MSE_loss_ang", "mse_ang"), ("train_cos_sim_lin", "cos_sim_lin"),
                      ("train_avg_cos_sim", "avg_cos_sim"), ("train_RMSE_loss", "rmse")):
        assert abs(float(w.logged[key]) - o[okey]) <= 1e-9 * max(1.0, abs(o[okey])), key
    with pytest.raises(ValueError):
        wrappers.COM_Base_Lightning("adam", 1e-3)


@pytest.mark.gpu
def test_flat_adam_follows_torch_adam_and_round_trips_its_state():
    """configure_optimizers returns FlatAdam (a torch.optim.Adam; one mshgnn_adam_step launch per step on the flat buffers): three steps
    track torch.optim.Adam on a twin model; its state_dict loads into torch's Adam and back, and both continue in step."""
    from morphsym_hgnn_amd.optim import FlatAdam
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    dev = torch.device("cuda")
    batch = _batch(x_dict, ei, y, case["B"], dev)
    twins = []
    for _ in range(2):
        w = _wrapper(case, spec, batch).to(dev)
        w.model.load_state_dict(params)
        w.lr = 1e-3
        twins.append(w)
    a, b = twins
    oa = a.configure_optimizers()
    ob = torch.optim.Adam(b.parameters(), lr=1e-3)
    assert isinstance(oa, FlatAdam) and isinstance(oa, torch.optim.Adam)

    def step(w, opt):
        loss = w.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    def worst():
        return max(float((p.detach() - q.detach()).abs().max() / q.detach().abs().max().clamp_min(1e-12))
                   for p, q in zip(a.model.parameters(), b.model.parameters()))
    for _ in range(3):
        step(a, oa); step(b, ob)
    assert oa._t == 3 and worst() < 5e-5            # (fp32 master weights on both sides; the two Adams differ in rounding only, which the
                                                    #  next step's gradients amplify a little)
    # state round trip: FlatAdam -> torch Adam on the twin, torch Adam -> FlatAdam, then one more step each
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert int(sd_a["state"][0]["step"]) == 3 and sd_a["state"][0]["exp_avg"].shape == sd_b["state"][0]["exp_avg"].shape
    ob.load_state_dict(sd_a); oa.load_state_dict(sd_b)
    step(a, oa); step(b, ob)
    assert oa._t == 4 and worst() < 1e-4            # (a fourth step on the swapped states: the rounding differences of three updates, amplified once more --
                                                    #  5.8e-5 measured with explicitly fused multiply-adds in the decoder tail, 4e-5 before)
    # gradients that are not the flat views (here: cleared) -> torch's own step, nothing breaks
    oa.zero_grad(set_to_none=True)
    oa.step()


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["bf16", "x3"])
@pytest.mark.parametrize("kind", ["a1c2_regression", "mck4_classification"])
def test_window_batch_training_step_gathers_in_the_encoder_and_matches_the_assembled_batch(kind, plan):
    """A `windows.WindowBatch` (window indices of a resident sequence) through `training_step`: the encoder gathers its inputs from the
    series (mshgnn_step_mse_series / _ce_series); loss and every gradient are the bits of the same step on the assembled batch, and the
    batch still serves `x_dict` / `y` to consumers that want tensors (validation)."""
    import types
    from morphsym_hgnn_amd import wrappers
    from morphsym_hgnn_amd.windows import SequenceStore, WindowBatch, quadsdk_a1_c2_recipe, minicheetah_k4_recipe
    from tests import test_windows as tw
    torch.set_default_dtype(torch.float32)
    dev = torch.device("cuda")
    if kind.startswith("a1c2"):
        seq, n, recipe = tw.SEQ, tw.N, quadsdk_a1_c2_recipe(tw.JP, tw.FP, tw.T, 3)
        spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
        _, cfg = helpers.load_group("a1-c2")
        make = lambda dummy: wrappers.HGNN_C2_Lightning_Reg(128, 2, spec.topology.metadata(), dummy, symmetry_mode="MorphSym", group_operator_path=cfg)
    else:
        seq, n, recipe = dict(tw.SEQ4), int(tw.FX4["N"]), minicheetah_k4_recipe(tw.JP, tw.FP, tw.T)
        spec = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 2, regression=False)
        _, cfg = helpers.load_group("mini_cheetah-k4")
        make = lambda dummy: wrappers.HGNN_K4_Lightning(128, 2, spec.topology.metadata(), dummy, regression=False, symmetry_mode="MorphSym",
                                                        group_operator_path=cfg)
    store = SequenceStore(seq, recipe, dtype=plan)
    B = 96
    starts = torch.randint(0, n - tw.T + 1, (B,), generator=torch.Generator().manual_seed(5))
    ei = spec.topology.edge_index_dict(B, device=dev)
    xs, y, _ = store.assemble(starts)
    plain = types.SimpleNamespace(x_dict={t: x.clone() for t, x in zip(recipe.node_types, xs)}, edge_index_dict=ei, y=y.clone(), batch_size=B)
    dummy = types.SimpleNamespace(edge_index_dict=ei, x_dict={t: x[:, :recipe.width(t)].float().contiguous() for t, x in plain.x_dict.items()})
    import os
    prev = os.environ.get("MSHGNN_DTYPE")
    os.environ["MSHGNN_DTYPE"] = plan
    try:
        torch.manual_seed(3)
        w = make(dummy).to(dev)
    finally:
        os.environ.pop("MSHGNN_DTYPE", None) if prev is None else os.environ.__setitem__("MSHGNN_DTYPE", prev)
    loss_a = w.training_step(plain, 0)
    loss_a.backward()
    g_a = w.model._gflat.clone(); la = loss_a.detach().clone()
    w.model.zero_grad()
    wb = store.batch(starts, ei)
    assert isinstance(wb, WindowBatch) and wb._x is None and next(iter(w.model._engines.values())).storage == plan
    loss_b = w.training_step(wb, 1)
    assert wb._x is not None and torch.equal(wb.y, y.to(dev))          # the step left the materialised windows and the labels on the batch
    loss_b.backward()
    torch.cuda.synchronize()
    assert torch.equal(la, loss_b.detach()) and torch.equal(g_a, w.model._gflat)
    for t in recipe.node_types:
        assert torch.equal(wb.x_dict[t], plain.x_dict[t])
    # a fresh WindowBatch without the fused route (validation under no_grad): tensors assembled on first access
    with torch.no_grad():
        v = w.validation_step(store.batch(starts, ei), 0)
    assert abs(float(v) - float(la)) <= 1e-5 * abs(float(la))


@pytest.mark.gpu
def test_wrapper_state_dict_is_the_lightning_checkpoint_layout():
    """A wrapper's state_dict has the keys of the reference's Lightning checkpoints (`model.<parameter>`, gnnLightning.py:962-984): the
    `state_dict` of a checkpoint written by `checkpoint.to_lightning_checkpoint` loads into it directly and reproduces the golden output."""
    from morphsym_hgnn_amd import checkpoint
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    dev = torch.device("cuda")
    batch = _batch(x_dict, ei, y, case["B"], dev)
    w = _wrapper(case, spec, batch).to(dev)
    assert list(w.state_dict().keys()) == ["model." + k for k in spec.param_shapes().keys()]
    src = _wrapper(case, spec, batch)
    src.model.load_state_dict(params)
    ckpt = checkpoint.to_lightning_checkpoint(src.model)
    w.load_state_dict(ckpt["state_dict"])
    with torch.no_grad():
        out = w.model(x_dict=batch.x_dict, edge_index_dict=batch.edge_index_dict)
    ref = torch.from_numpy(fx["out"])
    assert float((out.double().cpu().reshape(ref.shape) - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.gpu
def test_graphed_training_step_replays_the_eager_step_bit_for_bit():
    """wrappers.GraphedTrainingStep: zero_grad + training_step + backward + FlatAdam(graph_safe).step captured once in a HIP graph.  Three batches through the
    replayed graph == the same three batches through the eager step on a twin (same kernels, same order: identical parameter bits, identical losses), and
    building the graph does not train (parameters, optimizer state untouched by its warm-up)."""
    from morphsym_hgnn_amd import wrappers
    from morphsym_hgnn_amd.optim import FlatAdam
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    dev = torch.device("cuda")
    B = case["B"]
    batches = []
    for s in range(4):
        g = torch.Generator().manual_seed(100 + s)
        xb = {k: torch.randn(v.shape, generator=g, dtype=torch.float64) * (0.0 if k == "foot" else 1.0) + (1.0 if k == "foot" else 0.0) for k, v in x_dict.items()}
        batches.append(_batch(xb, ei, torch.randn(y.shape, generator=g, dtype=torch.float64), B, dev))
    twins = []
    for _ in range(2):
        w = _wrapper(case, spec, batches[0]).to(dev)
        w.model.load_state_dict(params)
        w.lr = 1e-3
        w.graph_safe_optimizer = True
        twins.append((w, w.configure_optimizers()))
    (a, oa), (b, ob) = twins
    assert isinstance(oa, FlatAdam) and oa._graph_safe
    before = [p.detach().clone() for p in a.parameters()]
    gs = wrappers.GraphedTrainingStep(a, oa, batches[0])
    for p, q in zip(a.parameters(), before):
        assert torch.equal(p.detach().float(), q.float()), "building the graph must not train"      # (the first forward moves the parameters into the flat fp32 buffer)
    assert int(oa._t_dev.item()) == 0
    losses_a, losses_b = [], []
    for bt in batches[1:]:
        losses_a.append(float(gs(bt)))
        ob.zero_grad(set_to_none=True)
        l = b.training_step(bt, 0); l.backward(); ob.step()
        losses_b.append(float(l))
    torch.cuda.synchronize()
    assert losses_a == losses_b, (losses_a, losses_b)
    for p, q in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(p.detach(), q.detach())
    assert int(oa._t_dev.item()) == 3 and oa.state_dict()["state"][0]["step"] == 3
    # the device-counted Adam tracks torch.optim.Adam like the host-counted one (bias corrections from the device's powf)
    c = _wrapper(case, spec, batches[0]).to(dev)
    c.model.load_state_dict(params)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3)
    for bt in batches[1:]:
        oc.zero_grad(set_to_none=True)
        l = c.training_step(bt, 0); l.backward(); oc.step()
    worst = max(float((p.detach() - q.detach()).abs().max() / q.detach().abs().max().clamp_min(1e-12)) for p, q in zip(a.model.parameters(), c.model.parameters()))
    assert worst < 5e-5, worst
    with pytest.raises(ValueError, match="graph_safe"):
        wrappers.GraphedTrainingStep(c, FlatAdam(c.model, lr=1e-3), batches[0])


@pytest.mark.gpu
def test_a_forward_on_other_data_between_forward_and_backward_is_refused():
    """the encoder of the no-cast route writes the engine's own row buffers on EVERY forward of a batch size, training or not -- a no_grad forward
    on other data of the same size between a training forward and its backward must make that backward raise (it would read the other batch's rows for the
    encoder's weight gradients), and a materialised WideInputs is refused once its rows have been overwritten."""
    from morphsym_hgnn_amd import engine as eng, models, synth
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    dev = torch.device("cuda", 0)
    B = case["B"]
    e = eng.Engine(spec, "bf16", device=dev)
    flat = eng.flatten_params(spec, params, dev)
    x1 = {k: v.to(dev, torch.float64) for k, v in x_dict.items()}
    x2 = {k: (v * 0.5 + 0.25).to(dev, torch.float64) for k, v in x_dict.items()}
    xs1 = e.cast_inputs(x1)
    assert isinstance(xs1, eng.WideInputs)
    t0 = e.stash_ticket(B)
    e.forward(xs1, flat, B, training=True)
    t1 = e.stash_ticket(B)
    assert t1 > t0
    xs2 = e.cast_inputs(x2)
    e.forward(xs2, flat, B, training=False)      # evaluation forward on OTHER data, same batch size: rewrites the row buffers
    assert e.stash_ticket(B) > t1, "an evaluation forward that rewrites the input rows must invalidate pending backwards"
    with pytest.raises(RuntimeError, match="overwritten"):
        e.backward(xs1, flat, torch.zeros(B * e.n_out, spec.out_channels, dtype=torch.float32, device=dev), B)
    # two batches of one size cast before either runs: the second takes the cast pass (tensors of its own), so the first still means its own data
    xa, xb = e.cast_inputs(x1), e.cast_inputs(x2)
    assert isinstance(xa, eng.WideInputs) and not isinstance(xb, eng.WideInputs)
    oa = e.forward(xa, flat, B, training=False).clone()
    ob = e.forward(xb, flat, B, training=False).clone()
    ra = e.forward(e.cast_inputs(x1), flat, B, training=False)
    assert torch.equal(oa, ra) and not torch.equal(oa, ob)
