"""Metric bookkeeping: the oracle against the reference's own known answers
(tests/testGnnLightning.py:399-413, :427-446, :465-500 -- inputs and expected values typed in below as data), and the
device kernels (through the C-ABI, morphsym_hgnn_amd.metrics.StepMetrics) against both."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as mo

# --- known answers held by the reference's tests ------------------------------------------------------------------
P16_IN = [[0.9, 0.3, 0.8, 0.55], [0.5, 0.5, 0.5, 0.5]]                      # testGnnLightning.py:400-401
Y16_IN = [[1, 0, 1, 1], [0, 1, 1, 0]]                                      # :402-403
P16_OUT = [[0.0063, 0.0077, 0.0252, 0.0308, 0.0027, 0.0033, 0.0108, 0.0132,
            0.0567, 0.0693, 0.2268, 0.2772, 0.0243, 0.0297, 0.0972, 0.1188], [0.0625] * 16]   # :405-412
Y16_OUT = [[11], [6]]                                                       # :413
REG_PRED = [[1, 1, 14, -50], [13, 13, 3, 12], [-101, 12, -13, -31]]         # :427-429
REG_Y = [[12, 13, 114, 0], [1, 1, 0, 1], [-100, 11, -10, -3]]               # :430-432
CLS_PRED = [[0.1, 11, 100, 19, 0.12, 0.14, 15, 24.45], [15, 11, 19, 19, 0.9898, 0.14, -10000, 24.45],
            [0.1, 13, 100, 19, 0.12, -10, 15, -24.45], [15, 11, 200, 19, 0.9898, 0.14, -10000, 44.45],
            [-0.1, 11, 100, 19, 0.12, 0.14, 15, 24.45], [-15, 11, 19, 19, -0.9898, 0.14, -10000, 24.45],
            [-0.1, 13, 100, 19, 0.12, -10, 15, -24.45], [-15, 11, 200, 19, -0.9898, 0.14, -10000, 44.45]]   # :465-472
CLS_Y = [[1, 1, 1, 1], [1, 1, 0, 1], [0, 1, 1, 0], [0, 1, 0, 0], [0, 0, 1, 1], [1, 1, 1, 0], [1, 0, 0, 0], [1, 0, 0, 1]]   # :473-480
CLS_ACC, CLS_F1 = 0.125, [0.7272727272727272, 0.0, 0.75, 0.8]               # :496-500


def _ce_reference_recipe(pred, y):
    """The recipe the reference test itself uses for the expected CE (testGnnLightning.py:487-493)."""
    ls = torch.nn.functional.log_softmax(torch.tensor(pred, dtype=torch.float64).reshape(-1, 2), dim=1)
    lab = torch.tensor(y).reshape(-1)
    return float(-(ls[torch.arange(ls.shape[0]), lab]).sum() / ls.shape[0])


def test_oracle_16_class_conversion_known_answer():
    p, y = mo.conversion_16_class(P16_IN, np.array(Y16_IN))
    np.testing.assert_array_almost_equal(np.array(P16_OUT), p, 15)
    np.testing.assert_array_equal(np.array(Y16_OUT), y)


def test_oracle_regression_metrics_known_answer():
    mse, rmse, l1 = mo.regression_metrics(REG_Y, REG_PRED)
    d = np.array(REG_PRED, dtype=np.float64).ravel() - np.array(REG_Y, dtype=np.float64).ravel()
    assert mse == np.mean(d * d) and rmse == np.sqrt(np.mean(d * d)) and l1 == np.mean(np.abs(d))


def test_oracle_classification_metrics_known_answer():
    m = mo.classification_metrics(CLS_Y, CLS_PRED)
    np.testing.assert_almost_equal(m["ce"], _ce_reference_recipe(CLS_PRED, CLS_Y), 5)
    assert m["acc"] == CLS_ACC
    assert m["f1"] == CLS_F1            # bit-exact: same float64 operation order as BinaryF1Score.compute


def test_oracle_body_to_world_matches_scipy():
    from scipy.spatial.transform import Rotation          # what the reference calls (gnnLightning.py:671-672)
    rng = np.random.default_rng(0)
    q = rng.normal(size=(17, 4)); g = rng.normal(size=(17, 12)) * 30
    R = Rotation.from_quat(q).inv().as_matrix()
    want = (R @ g.reshape(17, 4, 3).transpose(0, 2, 1)).transpose(0, 2, 1).reshape(17, 12)
    np.testing.assert_allclose(mo.body_frame_to_world_frame(q, g), want, rtol=0, atol=1e-12)


# --- device path ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_regression_metrics_known_answer_and_epoch():
    from morphsym_hgnn_amd.metrics import StepMetrics
    m = StepMetrics(regression=True)
    y, p = torch.tensor(REG_Y, dtype=torch.int), torch.tensor(REG_PRED, dtype=torch.float64)
    m.calculate_losses_step(y, p)
    mse, rmse, l1 = mo.regression_metrics(REG_Y, REG_PRED)
    assert m.mse_loss.item() == mse and m.rmse_loss.item() == rmse and m.l1_loss.item() == l1      # integers: exact
    # a second, random step; the epoch value is the ratio of the accumulated sums (torchmetrics semantics)
    g = torch.Generator().manual_seed(1)
    y2, p2 = torch.randn(8192, 12, generator=g), torch.randn(8192, 12, generator=g)
    m.calculate_losses_step(y2, p2)
    s_mse, _, s_l1 = mo.regression_metrics(y2.numpy(), p2.numpy())
    assert abs(m.mse_loss.item() - s_mse) <= 1e-12 * s_mse and abs(m.l1_loss.item() - s_l1) <= 1e-12 * s_l1
    m.calculate_losses_epoch()
    ya = np.concatenate([np.array(REG_Y, dtype=np.float64).ravel(), y2.numpy().ravel().astype(np.float64)])
    pa = np.concatenate([np.array(REG_PRED, dtype=np.float64).ravel(), p2.numpy().ravel().astype(np.float64)])
    e_mse, e_rmse, e_l1 = mo.regression_metrics(ya, pa)
    assert abs(m.mse_loss.item() - e_mse) <= 1e-12 * e_mse and abs(m.rmse_loss.item() - e_rmse) <= 1e-12 * e_rmse
    assert abs(m.l1_loss.item() - e_l1) <= 1e-12 * e_l1
    m.reset_all_metrics()
    m.calculate_losses_step(y, p)
    m.calculate_losses_epoch()
    assert m.mse_loss.item() == mse


@pytest.mark.gpu
def test_device_classification_metrics_known_answer():
    from morphsym_hgnn_amd.metrics import StepMetrics
    m = StepMetrics(regression=False)
    m.calculate_losses_step(torch.tensor(CLS_Y, dtype=torch.int), torch.tensor(CLS_PRED, dtype=torch.float64))
    np.testing.assert_almost_equal(m.ce_loss.item(), _ce_reference_recipe(CLS_PRED, CLS_Y), 5)
    assert m.acc.item() == CLS_ACC
    assert [m.f1_leg0.item(), m.f1_leg1.item(), m.f1_leg2.item(), m.f1_leg3.item()] == CLS_F1
    # random logits at a realistic batch: integer counts bit-exact vs the oracle, CE to fp32-sum accuracy
    g = torch.Generator().manual_seed(2)
    lg = torch.randn(4096, 8, generator=g) * 3
    yy = (torch.rand(4096, 4, generator=g) > 0.5).int()
    m2 = StepMetrics(regression=False)
    m2.calculate_losses_step(yy, lg)
    o = mo.classification_metrics(yy.numpy(), lg.numpy())
    assert m2.acc.item() == o["acc"]
    assert [m2.f1_leg0.item(), m2.f1_leg1.item(), m2.f1_leg2.item(), m2.f1_leg3.item()] == o["f1"]
    assert abs(m2.ce_loss.item() - o["ce"]) <= 1e-6 * o["ce"]
    p16, y16 = StepMetrics.classification_conversion_16_class(torch.tensor(P16_IN, dtype=torch.float64), torch.tensor(Y16_IN))
    np.testing.assert_array_almost_equal(np.array(P16_OUT), p16.numpy(), 15)
    np.testing.assert_array_equal(np.array(Y16_OUT), y16.numpy())


@pytest.mark.gpu
def test_device_body_to_world_matches_oracle():
    from morphsym_hgnn_amd.metrics import StepMetrics
    rng = np.random.default_rng(3)
    q = rng.normal(size=(1000, 4)).astype(np.float32); g = (rng.normal(size=(1000, 12)) * 30).astype(np.float32)
    got = StepMetrics(True).body_frame_to_world_frame(torch.from_numpy(q), torch.from_numpy(g)).cpu().numpy()
    want = mo.body_frame_to_world_frame(q, g)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_step_loss_carries_autograd_regression(dtype):
    """The loss `training_step` returns (gnnLightning.py:709-722) is differentiable, as the reference's torchmetrics value is: value and
    dL/dy_pred against torch's own arithmetic in fp64."""
    from morphsym_hgnn_amd.metrics import StepMetrics
    g = torch.Generator().manual_seed(3)
    B = 517
    y_pred = (torch.randn(B, 12, generator=g, dtype=torch.float64) * 30).to(dtype).cuda().requires_grad_(True)
    y = (torch.randn(B, 12, generator=g, dtype=torch.float64) * 30).to(dtype).cuda()
    m = StepMetrics(regression=True)
    m.calculate_losses_step(y, y_pred)
    assert m.mse_loss.requires_grad and not m.rmse_loss.requires_grad
    (3.0 * m.mse_loss).backward()
    ref_in = y_pred.detach().double().cpu().requires_grad_(True)
    ref = ((ref_in.flatten() - y.double().cpu().flatten()) ** 2).mean()
    (3.0 * ref).backward()
    assert abs(float(m.mse_loss) - float(ref)) <= 1e-6 * float(ref)
    assert y_pred.grad.dtype == dtype and y_pred.grad.shape == y_pred.shape
    assert float((y_pred.grad.double().cpu() - ref_in.grad).abs().max()) <= 2e-6 * float(ref_in.grad.abs().max())
    # without autograd on the prediction the published values are plain tensors, as before; the multi-workgroup sums are bit-reproducible
    m.calculate_losses_step(y, y_pred.detach())
    assert not m.mse_loss.requires_grad
    first = (m.mse_loss.clone(), m.l1_loss.clone())
    for _ in range(5):
        m.calculate_losses_step(y, y_pred.detach())
        assert torch.equal(m.mse_loss, first[0]) and torch.equal(m.l1_loss, first[1])
    with torch.no_grad():
        m.calculate_losses_step(y, y_pred)
    assert not m.mse_loss.requires_grad


@pytest.mark.gpu
def test_step_loss_carries_autograd_classification():
    from morphsym_hgnn_amd.metrics import StepMetrics
    g = torch.Generator().manual_seed(4)
    B = 333
    y_pred = (torch.randn(B, 8, generator=g, dtype=torch.float64) * 3).cuda().requires_grad_(True)      # step_helper_function's [B, 2 * 4] logits
    y = torch.randint(0, 2, (B, 4), generator=g).cuda()
    m = StepMetrics(regression=False)
    m.calculate_losses_step(y, y_pred)
    assert m.ce_loss.requires_grad
    m.ce_loss.backward()
    ref_in = y_pred.detach().cpu().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in.reshape(B * 4, 2), y.cpu().long().flatten(), reduction="sum") / (B * 4)
    ref.backward()
    assert abs(float(m.ce_loss) - float(ref)) <= 1e-6 * float(ref)
    assert float((y_pred.grad.cpu() - ref_in.grad).abs().max()) <= 2e-6 * float(ref_in.grad.abs().max())
    assert 0.0 <= float(m.acc) <= 1.0


@pytest.mark.gpu
def test_accumulating_entry_points_agree_with_the_step_entry_points():
    """mshgnn_metrics_regression / _classification (one workgroup, ADD into a caller state) against the one-launch step entry points the
    StepMetrics class uses: same fp64 sums up to the order of the partial sums; integer counts identical."""
    import ctypes as C
    from morphsym_hgnn_amd import engine as eng
    from morphsym_hgnn_amd.metrics import StepMetrics
    lib = eng.load_library()
    g = torch.Generator().manual_seed(9)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p, y = torch.randn(5000, 12, generator=g).cuda(), torch.randn(5000, 12, generator=g).cuda()
    state = torch.zeros(3, dtype=torch.float64, device="cuda")
    for _ in range(2):
        eng._check(lib, lib.mshgnn_metrics_regression(p.data_ptr(), y.data_ptr(), p.numel(), state.data_ptr(), st), "mshgnn_metrics_regression")
    m = StepMetrics(regression=True)
    m.calculate_losses_step(y, p); m.calculate_losses_step(y, p)
    torch.cuda.synchronize()
    assert torch.allclose(state, m._epoch_f[:3], rtol=1e-13, atol=0) and float(state[2]) == 2 * p.numel()
    lg = (torch.randn(3000, 8, generator=g) * 3).cuda()
    yy = (torch.rand(3000, 4, generator=g) > 0.5).int().cuda()
    ce, cnt = torch.zeros(2, dtype=torch.float64, device="cuda"), torch.zeros(18, dtype=torch.int64, device="cuda")
    eng._check(lib, lib.mshgnn_metrics_classification(lg.data_ptr(), yy.data_ptr(), 3000, ce.data_ptr(), cnt.data_ptr(), st), "mshgnn_metrics_classification")
    mc = StepMetrics(regression=False)
    mc.calculate_losses_step(yy, lg)
    torch.cuda.synchronize()
    assert torch.equal(cnt, mc._epoch_i) and torch.allclose(ce, mc._epoch_f[:2], rtol=1e-13, atol=0)


# --- centroidal-momentum wrappers' metrics (gnnLightning_com.py:96-121) ----------------------------------------------------------------
def _com_case(B=257, nb=4, seed=11):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(B, nb * 6, generator=g, dtype=torch.float64)
    p = y + 0.3 * torch.randn(B, nb * 6, generator=g, dtype=torch.float64)
    mean = torch.randn(6, generator=g, dtype=torch.float64)
    std = torch.rand(6, generator=g, dtype=torch.float64) + 0.5
    return y, p, mean, std


def test_oracle_com_metrics_match_torch_cosine_similarity():
    """The reference's tests hold no known answer for CosineSimilarityMetric: the oracle is pinned against torch.nn.CosineSimilarity and
    plain torch arithmetic, following COM_Base_Lightning.calculate_losses_step line by line."""
    y, p, mean, std = _com_case()
    nb = 4
    o = mo.com_metrics(y.numpy(), p.numpy(), nb, mean.numpy(), std.numpy())
    yv, pv = y.view(-1, nb, 6), p.view(-1, nb, 6)
    assert abs(o["mse"] - float(((p - y) ** 2).mean())) < 1e-14
    assert abs(o["mse_lin"] - float(((pv[:, :, :3] - yv[:, :, :3]) ** 2).mean())) < 1e-14
    assert abs(o["mse_ang"] - float(((pv[:, :, 3:] - yv[:, :, 3:]) ** 2).mean())) < 1e-14
    yu, pu = yv * std + mean, pv * std + mean
    cs = torch.nn.CosineSimilarity(dim=1)
    assert abs(o["cos_sim_lin"] - float(cs(pu[:, 0, :3], yu[:, 0, :3]).sum() / y.shape[0])) < 1e-14
    assert abs(o["cos_sim_ang"] - float(cs(pu[:, 0, 3:], yu[:, 0, 3:]).sum() / y.shape[0])) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("B,nb", [(1, 1), (257, 4), (8192, 2)])
def test_device_com_metrics_match_oracle_and_accumulate(B, nb):
    from morphsym_hgnn_amd.metrics import ComStepMetrics
    y, p, mean, std = _com_case(B, nb)
    m = ComStepMetrics(nb, mean, std)
    pr = p.cuda().requires_grad_(True)
    m.calculate_losses_step(y.cuda(), pr)
    o = mo.com_metrics(y.float().numpy(), p.float().numpy(), nb, mean.numpy(), std.numpy())        # (the kernels read fp32 copies)
    for name, key in (("mse_loss", "mse"), ("rmse_loss", "rmse"), ("mse_loss_lin", "mse_lin"), ("mse_loss_ang", "mse_ang"),
                      ("cos_sim_lin", "cos_sim_lin"), ("cos_sim_ang", "cos_sim_ang"), ("avg_cos_sim", "avg_cos_sim"), ("loss", "mse")):
        assert abs(float(getattr(m, name)) - o[key]) <= 1e-12 * max(1.0, abs(o[key])), name
    assert m.loss.requires_grad
    m.loss.backward()
    ref = 2 * (p.float().double() - y.float().double()) / p.numel()
    assert float((pr.grad.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # epoch = ratio of accumulated sums; two different batches
    y2, p2, *_ = _com_case(B, nb, seed=12)
    m.calculate_losses_step(y2.cuda(), p2.cuda())
    m.calculate_losses_epoch()
    oa = mo.com_metrics(torch.cat([y, y2]).float().numpy(), torch.cat([p, p2]).float().numpy(), nb, mean.numpy(), std.numpy())
    assert abs(float(m.cos_sim_lin) - oa["cos_sim_lin"]) <= 1e-12 and abs(float(m.mse_loss_ang) - oa["mse_ang"]) <= 1e-12 * oa["mse_ang"]
    m.reset_all_metrics()
    m.calculate_losses_step(y.cuda(), p.cuda()); m.calculate_losses_epoch()
    assert abs(float(m.mse_loss_lin) - o["mse_lin"]) <= 1e-12 * o["mse_lin"]
