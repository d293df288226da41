"""ORACLE (test infrastructure only -- imported by tests/ and nothing else): float64 CPU restatement of the metric
bookkeeping of the reference's Lightning wrappers, torchmetrics-free.

Restates src/ms_hgnn/lightning_py/gnnLightning.py:124-151 (calculate_losses_step), :285-348 (softmax helper, 16-class
conversion), :663-676 (body_frame_to_world_frame), customMetrics.py:5-54 (CrossEntropyLossMetric, BinaryF1Score) and, for the
centroidal-momentum wrappers, gnnLightning_com.py:96-121 with customMetrics.py:56-95 (CosineSimilarityMetric; pinned against
torch.nn.CosineSimilarity itself in tests/test_metrics.py -- the reference's tests hold no known answer for it).
torchmetrics 1.x (environment_files/requirements.txt) is absent from this image; its three members used here have
published closed forms: MeanSquaredError(squared=True/False) = sum sq err / n (sqrt of it), MeanAbsoluteError =
sum |err| / n, multiclass Accuracy(micro) = correct / total.

Pinned by the reference's own known answers, typed in as data in tests/test_metrics.py:
tests/testGnnLightning.py:399-413 (16-class conversion), :427-446 (MSE / RMSE / L1), :465-500 (CE, acc 0.125, F1 per leg).
"""
import numpy as np


def regression_metrics(y, y_pred):
    """gnnLightning.py:124-130 -> (mse, rmse, l1)."""
    y = np.asarray(y, dtype=np.float64).ravel()
    p = np.asarray(y_pred, dtype=np.float64).ravel()
    mse = np.mean(np.square(p - y))
    return mse, np.sqrt(mse), np.mean(np.abs(p - y))


def classification_useful_values(y_pred, batch_size):
    """gnnLightning.py:285-304: logits [B*4, 2], softmax probabilities, contact probability [B, 4]."""
    lg = np.asarray(y_pred, dtype=np.float64).reshape(batch_size * 4, 2)
    e = np.exp(lg - lg.max(axis=1, keepdims=True))
    prob = e / e.sum(axis=1, keepdims=True)
    return lg, prob, prob[:, 1].reshape(batch_size, 4)


def conversion_16_class(p1, y):
    """gnnLightning.py:306-348: labels -> 8 y0 + 4 y1 + 2 y2 + y3; probabilities -> products, class j = contact bits of j."""
    p1 = np.asarray(p1, dtype=np.float64)
    y = np.asarray(y)
    y_new = (y[:, 0] * 8 + y[:, 1] * 4 + y[:, 2] * 2 + y[:, 3]).reshape(-1, 1).astype(np.int64)
    out = np.zeros((p1.shape[0], 16))
    for j in range(16):
        f0 = p1[:, 0] if (j // 8) % 2 else 1 - p1[:, 0]
        f1 = p1[:, 1] if (j // 4) % 2 else 1 - p1[:, 1]
        f2 = p1[:, 2] if (j // 2) % 2 else 1 - p1[:, 2]
        f3 = p1[:, 3] if j % 2 else 1 - p1[:, 3]
        out[:, j] = (f0 * f1) * (f2 * f3)
    return out, y_new


def f1_from_counts(tp, fp, fn):
    """customMetrics.py:51-54 (same operation order; 0/0 -> 0 as nan_to_num does)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.float64(tp) / np.float64(tp + fp)
        recall = np.float64(tp) / np.float64(tp + fn)
        v = 2 * (precision * recall) / (precision + recall)
    return 0.0 if np.isnan(v) else float(v)


def classification_metrics(y, y_pred):
    """gnnLightning.py:132-151 -> dict(ce, acc, f1 = [4], counts = [4][tp, fp, fn, tn])."""
    y = np.asarray(y).astype(np.int64)
    B = np.asarray(y_pred).shape[0]
    lg, prob, p1 = classification_useful_values(y_pred, B)
    lab = y.reshape(-1)
    lse = np.log(np.exp(lg - lg.max(axis=1, keepdims=True)).sum(axis=1)) + lg.max(axis=1)
    ce = float(np.float32((lse - lg[np.arange(4 * B), lab]).sum())) / (4 * B)      # summed_loss.float() / total_num
    p16, y16 = conversion_16_class(p1, y)
    acc = float((np.argmax(p16, axis=1) == y16[:, 0]).mean())
    pred2 = np.argmax(prob, axis=1).reshape(B, 4)
    f1, counts = [], []
    for k in range(4):
        tp = int(((pred2[:, k] == 1) & (y[:, k] == 1)).sum()); fp = int(((pred2[:, k] == 1) & (y[:, k] == 0)).sum())
        fn = int(((pred2[:, k] == 0) & (y[:, k] == 1)).sum()); tn = int(((pred2[:, k] == 0) & (y[:, k] == 0)).sum())
        counts.append([tp, fp, fn, tn]); f1.append(f1_from_counts(tp, fp, fn))
    return {"ce": ce, "acc": acc, "f1": f1, "counts": counts}


def body_frame_to_world_frame(quat, grf_body):
    """gnnLightning.py:663-676, closed form: R(q) of the normalised scalar-last quaternion, world = R^T f per foot
    (tests cross-check this against scipy.spatial.transform.Rotation, which the reference calls)."""
    q = np.asarray(quat, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)   # [N, 3, 3]
    f = np.asarray(grf_body, dtype=np.float64).reshape(q.shape[0], 4, 3)
    return np.einsum("nji,nfj->nfi", R, f).reshape(q.shape[0], 12)


def com_metrics(y, y_pred, n_bases, y_mean, y_std, eps=1e-8):
    """COM_Base_Lightning.calculate_losses_step (gnnLightning_com.py:96-121) -> dict(mse, rmse, mse_lin, mse_ang, cos_sim_lin,
    cos_sim_ang, avg_cos_sim).  y / y_pred: [B, n_bases * 6] standardised; the cosine similarities are taken on base node 0 after
    un-standardising (soloDataset.py:33-46) with torch.nn.CosineSimilarity(dim=1)'s definition, batch mean (customMetrics.py:56-95)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1, n_bases, 6)
    p = np.asarray(y_pred, dtype=np.float64).reshape(-1, n_bases, 6)
    mse = np.mean(np.square(p - y))
    out = {"mse": mse, "rmse": np.sqrt(mse), "mse_lin": np.mean(np.square(p[:, :, :3] - y[:, :, :3])),
           "mse_ang": np.mean(np.square(p[:, :, 3:] - y[:, :, 3:]))}
    yu = y * np.asarray(y_std, dtype=np.float64) + np.asarray(y_mean, dtype=np.float64)
    pu = p * np.asarray(y_std, dtype=np.float64) + np.asarray(y_mean, dtype=np.float64)

    def cos(a, b):
        an = np.maximum(np.linalg.norm(a, axis=1, keepdims=True), eps)
        bn = np.maximum(np.linalg.norm(b, axis=1, keepdims=True), eps)
        return np.mean(np.sum((a / an) * (b / bn), axis=1))
    out["cos_sim_lin"] = cos(pu[:, 0, :3], yu[:, 0, :3])
    out["cos_sim_ang"] = cos(pu[:, 0, 3:], yu[:, 0, 3:])
    out["avg_cos_sim"] = (out["cos_sim_lin"] + out["cos_sim_ang"]) / 2
    return out
