"""Step / epoch metrics of the reference's Lightning wrappers, computed on the MI355X through the C-ABI.

Mirrors the metric half of `Base_Lightning` (src/ms_hgnn/lightning_py/gnnLightning.py): `calculate_losses_step`
(:124-151, regression MSE / RMSE / L1; classification CE, 16-class accuracy, per-leg F1), `calculate_losses_epoch`
(:153-164), `reset_all_metrics` (:166-177), `classification_conversion_16_class` (:306-348) and
`body_frame_to_world_frame` (:663-676, which in the reference hops to the CPU and scipy every step).  The reference's
torchmetrics / customMetrics states are plain sums; here they live in two small device buffers that the kernels
`mshgnn_metrics_regression` / `mshgnn_metrics_classification` add into (include/mshgnn.h).  Same attribute names as the
reference (`mse_loss`, `rmse_loss`, `l1_loss`, `ce_loss`, `acc`, `f1_leg0..3`), values are float64 tensors.  As in the reference, the
loss a `training_step` returns (`mse_loss` / `ce_loss`, gnnLightning.py:709-722) carries autograd when `y_pred` does: its backward is one
launch of `mshgnn_mse_loss` / `mshgnn_ce_loss` (dL/dy_pred on the device), so a wrapper built on this class trains without torch's
elementwise loss kernels.

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


class _StepLoss(torch.autograd.Function):
    """The batch loss of `calculate_losses_step` as an autograd node.  forward: ONE launch -- this step's metric sums, the loss, their accumulation
    into the epoch state and dL/dy_pred (2 (y_pred - y) / n, or (softmax - onehot) / rows); backward: that gradient times the upstream scalar."""

    @staticmethod
    def forward(ctx, y_pred, y, metrics):
        buf, g = metrics._launch(y, y_pred, True)
        metrics._cur = buf          # (the sums are no autograd outputs: the only output is the loss the kernel left next to them)
        ctx.g, ctx.meta = g, (y_pred.shape, y_pred.dtype, y_pred.device)
        return buf[:8].view(torch.float64)[3 if metrics.regression else 2]

    @staticmethod
    def backward(ctx, gl):
        shape, dtype, in_dev = ctx.meta
        g = ctx.g * gl.to(ctx.g.device)           # (a 0-dim factor does not promote the fp32 gradient; out of place: backward may run twice)
        return g.to(device=in_dev, dtype=dtype).view(shape), None, None


_REG_NAMES = ("mse_loss", "rmse_loss", "l1_loss")
_CLS_NAMES = ("ce_loss", "acc", "f1_leg0", "f1_leg1", "f1_leg2", "f1_leg3")


class StepMetrics:
    """Drop-in for the metric bookkeeping of `Base_Lightning` (`regression=True`: GRF / COM regression wrappers,
    `False`: contact classification).  The published values (`mse_loss`, ..., `f1_leg3`) are evaluated from the device sums when they are
    read, so a step that only needs its loss launches nothing for the others."""

    def __init__(self, regression: bool = True, device=None):
        self.regression = regression
        self.device = _device(device)
        self.lib = eng.load_library()
        # state layout (one buffer of 26 eight-byte words): [0:8] float64 -- regression: sq, abs, n | classification: ce, rows (a step's buffer also
        # holds the step's published values behind them, include/mshgnn.h); [8:26] int64 counts
        self._epoch = torch.zeros(26, dtype=torch.int64, device=self.device)
        self._scratch = torch.zeros(16384 // 8, dtype=torch.int64, device=self.device)      # MSHGNN_METRICS_SCRATCH_BYTES: partials + ticket
        self._cur = None          # the state the published values are read from: the last step's, or the epoch's
        self._loss = None         # the last step's loss when it carries autograd
        self._vals = {}
        self._from_step = False

    @property
    def _epoch_f(self):
        return self._epoch[:8].view(torch.float64)

    @property
    def _epoch_i(self):
        return self._epoch[8:]

    # ---- values from a state -----------------------------------------------------------------------------------
    def _names(self):
        return _REG_NAMES if self.regression else _CLS_NAMES

    def _value(self, name):
        if name not in self._names():
            raise AttributeError(f"'{name}' is not a metric of a {'regression' if self.regression else 'classification'} wrapper")
        if self._cur is None:
            return None
        if name in self._vals:
            return self._vals[name]
        f, i = self._cur[:8].view(torch.float64), self._cur[8:]
        if name in ("mse_loss", "ce_loss") and self._loss is not None:
            v = self._loss
        elif self._from_step:                              # the kernel left the step's values next to its sums: no launch
            v = f[(_REG_NAMES.index(name) + 3) if self.regression else (_CLS_NAMES.index(name) + 2)]
        elif name == "mse_loss":
            v = f[0] / f[2]
        elif name == "rmse_loss":
            v = torch.sqrt(f[0] / f[2])
        elif name == "l1_loss":
            v = f[1] / f[2]
        elif name == "ce_loss":
            v = f[0].float().double() / f[1]            # `summed_loss.float() / total_num`, customMetrics.py:24
        elif name == "acc":
            v = i[1].double() / i[0].double()
        else:
            k = int(name[-1])
            v = _f1(i[2 + 4 * k], i[3 + 4 * k], i[4 + 4 * k])
        self._vals[name] = v
        return v

    mse_loss = property(lambda self: self._value("mse_loss"))
    rmse_loss = property(lambda self: self._value("rmse_loss"))
    l1_loss = property(lambda self: self._value("l1_loss"))
    ce_loss = property(lambda self: self._value("ce_loss"))
    acc = property(lambda self: self._value("acc"))
    f1_leg0 = property(lambda self: self._value("f1_leg0"))
    f1_leg1 = property(lambda self: self._value("f1_leg1"))
    f1_leg2 = property(lambda self: self._value("f1_leg2"))
    f1_leg3 = property(lambda self: self._value("f1_leg3"))

    # ---- reference API -----------------------------------------------------------------------------------------
    def _launch(self, y: torch.Tensor, y_pred: torch.Tensor, want_grad: bool):
        """One launch: this batch's sums into a fresh state buffer, added into the epoch state, + dL/dy_pred (fp32) when asked for."""
        dev = self.device
        buf = torch.empty(26, dtype=torch.int64, device=dev)
        ep = self._epoch.data_ptr()
        with torch.cuda.device(dev):
            if self.regression:
                yp = y_pred.detach().to(dev, torch.float32).flatten().contiguous()
                yy = y.detach().to(dev, torch.float32).flatten().contiguous()
                if yp.numel() != yy.numel():
                    raise ValueError("y and y_pred must have the same number of elements")
                g = torch.empty_like(yp) if want_grad else None
                eng._check(self.lib, self.lib.mshgnn_metrics_regression_step(yp.data_ptr(), yy.data_ptr(), yp.numel(), buf.data_ptr(), ep,
                                                                             g.data_ptr() if want_grad else None, self._scratch.data_ptr(), _stream(dev)),
                           "mshgnn_metrics_regression_step")
            else:
                batch = y_pred.shape[0]
                yp = y_pred.detach().to(dev, torch.float32).reshape(batch * 4, 2).contiguous()     # gnnLightning.py:300
                yy = y.detach().to(dev, torch.int32).reshape(batch * 4).contiguous()
                g = torch.empty_like(yp) if want_grad else None
                eng._check(self.lib, self.lib.mshgnn_metrics_classification_step(yp.data_ptr(), yy.data_ptr(), batch, buf.data_ptr(), buf.data_ptr() + 64,
                                                                                 ep, ep + 64, g.data_ptr() if want_grad else None, self._scratch.data_ptr(),
                                                                                 _stream(dev)),
                           "mshgnn_metrics_classification_step")
        return buf, g

    def calculate_losses_step(self, y: torch.Tensor, y_pred: torch.Tensor):
        """Metrics of this batch (published as attributes) and accumulation into the epoch state.  When `y_pred` carries autograd, so does
        the published `mse_loss` / `ce_loss` -- what the reference's `training_step` returns for backward (gnnLightning.py:709-722)."""
        if torch.is_grad_enabled() and y_pred.requires_grad:
            self._loss = _StepLoss.apply(y_pred, y, self)
        else:
            self._loss = None
            self._cur, _ = self._launch(y, y_pred, False)
        self._vals, self._from_step = {}, True

    def set_step_loss(self, loss: torch.Tensor) -> None:
        """Publish `loss` (a differentiable scalar the caller computed for this step, e.g. the fused engine step's) as `mse_loss` / `ce_loss`."""
        self._loss = loss
        self._vals.pop("mse_loss" if self.regression else "ce_loss", None)

    def calculate_losses_epoch(self) -> None:
        self._cur, self._loss, self._vals, self._from_step = self._epoch.clone(), None, {}, False

    def reset_all_metrics(self) -> None:
        self._epoch.zero_()

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


class ComStepMetrics(StepMetrics):
    """Metric bookkeeping of the centroidal-momentum wrappers (`COM_Base_Lightning`, gnnLightning_com.py:28-232): the regression values of
    `StepMetrics` (`mse_loss` differentiable, `rmse_loss`) plus `mse_loss_lin`, `mse_loss_ang`, `cos_sim_lin`, `cos_sim_ang`, `avg_cos_sim`
    and `loss` (= `mse_loss`, :121).  `y_mean` / `y_std`: the 6 label statistics of the dataset's `Standarizer` (soloDataset.py:12-46) --
    the cosine similarities are taken on base node 0 after un-standardising.  Two launches per step."""

    def __init__(self, num_bases: int, y_mean, y_std, device=None):
        super().__init__(regression=True, device=device)
        self.num_bases = int(num_bases)
        mean = [float(v) for v in torch.as_tensor(y_mean, dtype=torch.float64).flatten().tolist()]
        std = [float(v) for v in torch.as_tensor(y_std, dtype=torch.float64).flatten().tolist()]
        if len(mean) != 6 or len(std) != 6:
            raise ValueError("y_mean and y_std must have 6 entries (lin(3) | ang(3))")
        self._mean, self._std = (C.c_double * 6)(*mean), (C.c_double * 6)(*std)
        self._com_epoch = torch.zeros(8, dtype=torch.float64, device=self.device)
        self._com_cur = None
        self._com_scratch = torch.zeros(16384 // 8, dtype=torch.int64, device=self.device)

    def calculate_losses_step(self, y: torch.Tensor, y_pred: torch.Tensor):
        super().calculate_losses_step(y, y_pred)
        dev = self.device
        yp = y_pred.detach().to(dev, torch.float32).flatten().contiguous()
        yy = y.detach().to(dev, torch.float32).flatten().contiguous()
        per = self.num_bases * 6
        if yp.numel() != yy.numel() or yp.numel() % per:
            raise ValueError(f"y and y_pred must hold batch x {self.num_bases} x 6 values")
        cur = torch.empty(8, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            eng._check(self.lib, self.lib.mshgnn_metrics_com_step(yp.data_ptr(), yy.data_ptr(), yp.numel() // per, self.num_bases, self._mean, self._std,
                                                                  cur.data_ptr(), self._com_epoch.data_ptr(), self._com_scratch.data_ptr(), _stream(dev)),
                       "mshgnn_metrics_com_step")
        self._com_cur = cur

    def calculate_losses_epoch(self) -> None:
        super().calculate_losses_epoch()
        self._com_cur = self._com_epoch.clone()

    def reset_all_metrics(self) -> None:
        super().reset_all_metrics()
        self._com_epoch.zero_()

    def _com(self, i, j):
        return None if self._com_cur is None else self._com_cur[i] / self._com_cur[j]

    mse_loss_lin = property(lambda self: self._com(0, 2))
    mse_loss_ang = property(lambda self: self._com(1, 3))
    cos_sim_lin = property(lambda self: self._com(4, 6))
    cos_sim_ang = property(lambda self: self._com(5, 6))
    avg_cos_sim = property(lambda self: None if self._com_cur is None else (self.cos_sim_lin + self.cos_sim_ang) / 2)
    loss = property(lambda self: self.mse_loss)
