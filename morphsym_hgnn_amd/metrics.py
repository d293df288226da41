"""Step / epoch metrics of the reference's Lightning wrappers, computed on the MI355X through the C-ABI.

Mirrors the metric half of `Base_Lightning` (src/ms_hgnn/lightning_py/gnnLightning.py): `calculate_losses_step`
(:124-151, regression MSE / RMSE / L1; classification CE, 16-class accuracy, per-leg F1), `calculate_losses_epoch`
(:153-164), `reset_all_metrics` (:166-177), `classification_conversion_16_class` (:306-348) and
`body_frame_to_world_frame` (:663-676, which in the reference hops to the CPU and scipy every step).  The reference's
torchmetrics / customMetrics states are plain sums; here they live in two small device buffers that the kernels
`mshgnn_metrics_regression` / `mshgnn_metrics_classification` add into (include/mshgnn.h).  Same attribute names as the
reference (`mse_loss`, `rmse_loss`, `l1_loss`, `ce_loss`, `acc`, `f1_leg0..3`), values are float64 tensors.

There is no CPU implementation: without a HIP device every method raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import engine as eng


def _device(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("the MS-HGNN metrics run on a HIP device; there is no CPU fallback")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f1(tp, fp, fn):
    """BinaryF1Score.compute (customMetrics.py:51-54), same operation order, float64; 0/0 -> 0 (nan_to_num)."""
    tp, fp, fn = tp.double(), fp.double(), fn.double()
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return torch.nan_to_num(2 * (precision * recall) / (precision + recall))


class StepMetrics:
    """Drop-in for the metric bookkeeping of `Base_Lightning` (`regression=True`: GRF / COM regression wrappers,
    `False`: contact classification)."""

    def __init__(self, regression: bool = True, device=None):
        self.regression = regression
        self.device = _device(device)
        self.lib = eng.load_library()
        self._epoch_f = torch.zeros(4, dtype=torch.float64, device=self.device)    # regression: sq, abs, n | classification: ce, rows
        self._epoch_i = torch.zeros(18, dtype=torch.int64, device=self.device)
        self.mse_loss = self.rmse_loss = self.l1_loss = None
        self.ce_loss = self.acc = self.f1_leg0 = self.f1_leg1 = self.f1_leg2 = self.f1_leg3 = None

    # ---- values from a state -----------------------------------------------------------------------------------
    def _publish(self, f, i):
        if self.regression:
            self.mse_loss = f[0] / f[2]
            self.rmse_loss = torch.sqrt(f[0] / f[2])
            self.l1_loss = f[1] / f[2]
        else:
            self.ce_loss = f[0].float().double() / f[1]            # `summed_loss.float() / total_num`, customMetrics.py:24
            self.acc = i[1].double() / i[0].double()
            for k in range(4):
                setattr(self, f"f1_leg{k}", _f1(i[2 + 4 * k], i[3 + 4 * k], i[4 + 4 * k]))

    # ---- reference API -----------------------------------------------------------------------------------------
    def calculate_losses_step(self, y: torch.Tensor, y_pred: torch.Tensor):
        """Metrics of this batch (published as attributes) and accumulation into the epoch state."""
        dev = self.device
        f = torch.zeros(4, dtype=torch.float64, device=dev)
        i = torch.zeros(18, dtype=torch.int64, device=dev)
        if self.regression:
            yp = y_pred.detach().to(dev, torch.float32).flatten().contiguous()
            yy = y.detach().to(dev, torch.float32).flatten().contiguous()
            if yp.numel() != yy.numel():
                raise ValueError("y and y_pred must have the same number of elements")
            eng._check(self.lib, self.lib.mshgnn_metrics_regression(yp.data_ptr(), yy.data_ptr(), yp.numel(), f.data_ptr(), _stream(dev)),
                       "mshgnn_metrics_regression")
        else:
            batch = y_pred.shape[0]
            yp = y_pred.detach().to(dev, torch.float32).reshape(batch * 4, 2).contiguous()     # gnnLightning.py:300
            yy = y.detach().to(dev, torch.int32).reshape(batch, 4).contiguous()
            eng._check(self.lib, self.lib.mshgnn_metrics_classification(yp.data_ptr(), yy.data_ptr(), batch, f.data_ptr(), i.data_ptr(),
                                                                        _stream(dev)), "mshgnn_metrics_classification")
        self._epoch_f += f
        self._epoch_i += i
        self._publish(f, i)

    def calculate_losses_epoch(self) -> None:
        self._publish(self._epoch_f, self._epoch_i)

    def reset_all_metrics(self) -> None:
        self._epoch_f.zero_()
        self._epoch_i.zero_()

    @staticmethod
    def classification_conversion_16_class(y_pred_per_foot_prob_only_1: torch.Tensor, y: torch.Tensor):
        """16-class probabilities and labels (gnnLightning.py:306-348) as tensors, vectorised: class j has bit (3 - k) of j
        set when foot k is in contact; P(j) = ((f0 f1)(f2 f3)) with f_k = p_k or 1 - p_k."""
        p = y_pred_per_foot_prob_only_1
        bits = torch.tensor([[(j >> (3 - k)) & 1 for k in range(4)] for j in range(16)], dtype=torch.bool, device=p.device)   # [16, 4]
        f = torch.where(bits.unsqueeze(0), p.unsqueeze(1), 1 - p.unsqueeze(1))                                               # [B, 16, 4]
        y_pred_new = (f[..., 0] * f[..., 1]) * (f[..., 2] * f[..., 3])
        w = torch.tensor([8, 4, 2, 1], dtype=torch.long, device=y.device)
        y_new = (y.long() * w).sum(dim=1, keepdim=True)
        return y_pred_new, y_new

    def body_frame_to_world_frame(self, batch_r_quat: torch.Tensor, grf_body_frame: torch.Tensor) -> torch.Tensor:
        """[N, 4] scalar-last world->body quaternions, [N, 12] body-frame GRFs -> [N, 12] world-frame GRFs, on device."""
        dev = self.device
        q = batch_r_quat.detach().to(dev, torch.float32).contiguous()
        g = grf_body_frame.detach().to(dev, torch.float32).contiguous()
        n = q.shape[0]
        if q.shape != (n, 4) or g.numel() != n * 12:
            raise ValueError("expected quaternions [N, 4] and 3-D GRFs [N, 12]")
        out = torch.empty(n, 12, dtype=torch.float32, device=dev)
        eng._check(self.lib, self.lib.mshgnn_grf_body_to_world(q.data_ptr(), g.data_ptr(), out.data_ptr(), n, _stream(dev)),
                   "mshgnn_grf_body_to_world")
        return out.to(grf_body_frame.dtype)
