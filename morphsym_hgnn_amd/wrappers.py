"""The training-step wrappers around the MS-HGNN models: host-side mirror of the reference's Lightning modules for the hot path.

Same class names, constructor arguments, method names and return values as `src/ms_hgnn/lightning_py/gnnLightning.py`:
`Base_Lightning` (:28-348), `Heterogeneous_GNN_Lightning` (:415-462), `HGNN_K4_Lightning` (:464-513), `HGNN_C2_Lightning_Cls`
(:515-562), `HGNN_C2_Lightning_Reg` (:564-778), and of `gnnLightning_com.py`: `COM_Base_Lightning` (:28-232), `COM_HGNN_Lightning`
(:290-340), `COM_HGNN_SYM_Lightning` (:343-409).  What differs is where the work runs: the model is this package's (HIP engine), the
metric bookkeeping is `metrics.StepMetrics` (one launch per step, the returned loss carries autograd) and the world-frame rotation of
the GRFs stays on the device instead of the reference's per-step CPU + scipy hop (:663-676).

`lightning` is not a dependency: with it installed the wrappers ARE `LightningModule`s and a `Trainer` drives them unchanged; without
it they are plain `nn.Module`s whose `log()` records the values in `self.logged`, and `examples/` / `tests/` drive the same methods
(`training_step`, `validation_step`, `on_validation_epoch_end`, `configure_optimizers`) by hand.  Datasets, checkpoint callbacks, W&B
logging and `train_model` / `evaluate_model` are out of scope (control plane).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn, optim

from . import models
from .metrics import StepMetrics

try:  # pragma: no cover - lightning is absent in the build image
    import lightning as _L
    _Base = _L.LightningModule
except Exception:  # noqa: BLE001
    _L = None
    _Base = nn.Module

_REG = ("MSE_loss", "mse_loss"), ("RMSE_loss", "rmse_loss"), ("L1_loss", "l1_loss")
_METRIC_ATTRS = ("mse_loss", "rmse_loss", "l1_loss", "ce_loss", "acc", "f1_leg0", "f1_leg1", "f1_leg2", "f1_leg3")


def _metric_property(name):
    def get(self):
        m = self.__dict__.get("_metrics")
        return None if m is None or name not in m._names() else getattr(m, name)
    return property(get)


def _model_fused_step(wrapper, batch):
    """(out, loss) from the model's one-call training step -- for a `windows.WindowBatch` with the window gather fused into the encoder --
    or None when the wrapper has to take the two-call route."""
    if not wrapper.fused_training_step:
        return None
    from .windows import WindowBatch
    if isinstance(batch, WindowBatch):
        r = wrapper.model.fused_training_step_windows(batch)
        if r is not None:
            return r
    return wrapper.model.fused_training_step(batch.x_dict, batch.edge_index_dict, batch.y)


class Base_Lightning(_Base):
    """Steps, epoch hooks, logging and the optimizer shared by every wrapper (gnnLightning.py:28-348)."""

    # training_step hands the labels to the model, which runs forward + loss + backward as ONE engine call (models.fused_training_step:
    # the decoder, the loss and the decoder's backward sit in the tail of the fused forward kernel); the returned loss delivers those
    # gradients when backward() is called on it.  Same loss and gradients as the two-call route, which is taken when this is False
    # (an attribute, set per wrapper or on the class), under torch.distributed, with host parameters, or without autograd.
    fused_training_step = True

    def __init__(self, optimizer: str, lr: float, regression: bool):
        super().__init__()
        self.optimizer = optimizer
        self.lr = lr
        self.regression = regression
        self.body_to_world_frame = False
        self.__dict__["_metrics"] = None            # metrics.StepMetrics, created with the first step (it needs the device)
        self.__dict__["_metrics_world"] = None
        self.logged = {}

    # ---- metric state ------------------------------------------------------------------------------------------------
    def _m(self, world: bool = False) -> StepMetrics:
        key = "_metrics_world" if world else "_metrics"
        if self.__dict__[key] is None:
            self.__dict__[key] = StepMetrics(regression=True if world else self.regression)
        return self.__dict__[key]

    if _L is None:
        def log(self, name, value, on_step: bool = False, on_epoch: bool = True, **_):
            """Stand-in for LightningModule.log: the last value per name (a device tensor; reading it is the caller's sync)."""
            self.logged[name] = value

    def log_losses(self, step_name: str, on_step: bool):
        on_epoch = not on_step
        if self.regression:
            for label, attr in _REG:
                self.log(f"{step_name}_{label}", getattr(self, attr), on_step=on_step, on_epoch=on_epoch)
        else:
            self.log(step_name + "_CE_loss", self.ce_loss, on_step=on_step, on_epoch=on_epoch)
            self.log(step_name + "_Accuracy", self.acc, on_step=on_step, on_epoch=on_epoch)
            f1 = [self.f1_leg0, self.f1_leg1, self.f1_leg2, self.f1_leg3]
            self.log(step_name + "_F1_Score_Leg_Avg", (f1[0] + f1[1] + f1[2] + f1[3]) / 4.0, on_step=on_step, on_epoch=on_epoch)
            for k in range(4):
                self.log(f"{step_name}_F1_Score_Leg_{k}", f1[k], on_step=on_step, on_epoch=on_epoch)

    def calculate_losses_step(self, y: torch.Tensor, y_pred: torch.Tensor):
        """gnnLightning.py:124-151.  Classification: `y_pred` [batch, 8] logits, `y` [batch, 4] contact flags."""
        self._m().calculate_losses_step(y, y_pred)

    def calculate_losses_epoch(self) -> None:
        self._m().calculate_losses_epoch()

    def reset_all_metrics(self) -> None:
        self._m().reset_all_metrics()

    def _loss(self):
        return self.mse_loss if self.regression else self.ce_loss

    # ---- steps (gnnLightning.py:179-256) -----------------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        y, y_pred = self.step_helper_function(batch)
        self.calculate_losses_step(y, y_pred)
        self.log_losses("train", on_step=True)
        return self._loss()

    def on_validation_epoch_start(self):
        self.reset_all_metrics()

    def validation_step(self, batch, batch_idx):
        y, y_pred = self.step_helper_function(batch)
        self.calculate_losses_step(y, y_pred)
        return self._loss()

    def on_validation_epoch_end(self):
        self.calculate_losses_epoch()
        self.log_losses("val", on_step=False)

    def on_test_epoch_start(self):
        self.reset_all_metrics()

    def test_step(self, batch, batch_idx):
        return self.validation_step(batch, batch_idx)

    def on_test_epoch_end(self):
        self.calculate_losses_epoch()
        self.log_losses("test", on_step=False)

    def on_predict_start(self):
        self.reset_all_metrics()

    def predict_step(self, batch, batch_idx):
        y, y_pred = self.step_helper_function(batch)
        self.calculate_losses_step(y, y_pred)
        if not self.regression:
            raise NotImplementedError("This prediction method is not fully tested for classification.")
        return y, y_pred

    def on_predict_end(self):
        self.calculate_losses_epoch()

    # ---- optimizer (gnnLightning.py:258-265) -------------------------------------------------------------------------
    def configure_optimizers(self):
        if self.optimizer == "adam":
            model = getattr(self, "model", None)
            if isinstance(model, models._MSHGNNBase) and len(list(self.parameters())) == len(list(model.parameters())):
                from .optim import FlatAdam      # a torch.optim.Adam whose step is one launch on the flat buffers (torch's own step otherwise)
                return FlatAdam(model, lr=self.lr, graph_safe=bool(getattr(self, "graph_safe_optimizer", False)))      # (graph_safe: GraphedTrainingStep)
            return optim.Adam(self.parameters(), lr=self.lr)
        if self.optimizer == "sgd":
            return optim.SGD(self.parameters(), lr=self.lr)
        raise ValueError("Invalid optimizer setting")

    # ---- helpers -----------------------------------------------------------------------------------------------------
    def step_helper_function(self, batch):
        raise NotImplementedError

    def classification_calculate_useful_values(self, y_pred, batch_size):
        """Per-foot logits [batch*4, 2], their softmax, and the contact probabilities [batch, 4] (gnnLightning.py:285-304)."""
        per_foot = torch.reshape(y_pred, (batch_size * 4, 2))
        prob = torch.nn.functional.softmax(per_foot, dim=1)
        return per_foot, prob, torch.reshape(prob[:, 1], (batch_size, 4))

    @staticmethod
    def classification_conversion_16_class(y_pred_per_foot_prob_only_1: torch.Tensor, y: torch.Tensor):
        return StepMetrics.classification_conversion_16_class(y_pred_per_foot_prob_only_1, y)


for _n in _METRIC_ATTRS:
    setattr(Base_Lightning, _n, _metric_property(_n))


class _HGNNWrapper(Base_Lightning):
    """What the four GRF wrappers share: build the model, run the lazy-initialising dummy forward (:445-447), reshape outputs and labels
    per window (:449-462, :495-513, :680-695)."""

    _label_width_is_output_width = False      # y: [batch, out_channels_per_foot * 4] (True) or [batch, 4] (False)

    def _finish_init(self, dummy_batch):
        with torch.no_grad():
            self.model(x_dict=dummy_batch.x_dict, edge_index_dict=dummy_batch.edge_index_dict)
        if _L is not None:  # pragma: no cover
            self.save_hyperparameters(ignore=["dummy_batch", "activation_fn"])

    def _shape(self, batch, out_raw):
        batch_size = batch.batch_size if hasattr(batch, "batch_size") else 1
        width = self.model.out_channels_per_foot * 4
        y_pred = torch.reshape(out_raw.squeeze(), (batch_size, width))
        y = torch.reshape(batch.y, (batch_size, width if self._label_width_is_output_width else 4))
        return y, y_pred

    def step_helper_function(self, batch):
        return self._shape(batch, self.model(x_dict=batch.x_dict, edge_index_dict=batch.edge_index_dict))

    def _fused_step(self, batch):
        """(y, y_pred, loss) from the one-call engine step, or None when that route does not apply."""
        r = _model_fused_step(self, batch)
        if r is None:
            return None
        y, y_pred = self._shape(batch, r[0])
        return y, y_pred, r[1]

    def training_step(self, batch, batch_idx):
        r = self._fused_step(batch)
        if r is None:
            return super().training_step(batch, batch_idx)
        y, y_pred, loss = r
        self.calculate_losses_step(y, y_pred)          # the step's metric sums (y_pred carries no autograd here)
        self._m().set_step_loss(loss)
        self.log_losses("train", on_step=True)
        return loss


class Heterogeneous_GNN_Lightning(_HGNNWrapper):
    """MI-HGNN baseline `GRF_HGNN` (gnnLightning.py:415-462)."""
    _label_width_is_output_width = True

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), grf_dimension: int = 1):
        super().__init__(optimizer, lr, regression)
        self.model = models.GRF_HGNN(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata,
                                     regression=regression, activation_fn=activation_fn, grf_dimension=grf_dimension)
        self._finish_init(dummy_batch)


class HGNN_K4_Lightning(_HGNNWrapper):
    """`GRF_HGNN_K4` (gnnLightning.py:464-513): labels [batch, 4] (1-D GRFs or contact flags)."""

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), symmetry_mode: Optional[str] = None,
                 group_operator_path: Optional[str] = None):
        super().__init__(optimizer, lr, regression)
        self.model = models.GRF_HGNN_K4(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata,
                                        regression=regression, activation_fn=activation_fn, symmetry_mode=symmetry_mode,
                                        group_operator_path=group_operator_path)
        self._finish_init(dummy_batch)


class HGNN_C2_Lightning_Cls(_HGNNWrapper):
    """`GRF_HGNN_C2` for contact classification (gnnLightning.py:515-562)."""

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), symmetry_mode: Optional[str] = None,
                 group_operator_path: Optional[str] = None):
        super().__init__(optimizer, lr, regression)
        self.model = models.GRF_HGNN_C2(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata,
                                        regression=regression, activation_fn=activation_fn, symmetry_mode=symmetry_mode,
                                        group_operator_path=group_operator_path)
        self._finish_init(dummy_batch)


class HGNN_C2_Lightning_Reg(_HGNNWrapper):
    """`GRF_HGNN_C2` for GRF regression, optionally with the metrics ALSO evaluated in the world frame (gnnLightning.py:564-778): with
    `grf_body_to_world_frame` the labels / predictions are rotated by the inverse of `batch.r_o` (world->body quaternions, scalar last)
    on the device and a second metric state accumulates them; the loss that is returned stays the body-frame MSE (:709-716)."""
    _label_width_is_output_width = True

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), symmetry_mode: Optional[str] = None,
                 group_operator_path: Optional[str] = None, grf_body_to_world_frame: Optional[bool] = None, grf_dimension: int = 3):
        super().__init__(optimizer, lr, regression)
        self.model = models.GRF_HGNN_C2(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata,
                                        regression=regression, activation_fn=activation_fn, symmetry_mode=symmetry_mode,
                                        group_operator_path=group_operator_path, grf_dimension=grf_dimension)
        self._finish_init(dummy_batch)
        self.body_to_world_frame = bool(grf_body_to_world_frame) if regression else False

    # world-frame values: read from the second metric state
    mse_loss_worldframe = property(lambda self: None if self.__dict__["_metrics_world"] is None else self._m(True).mse_loss)
    rmse_loss_worldframe = property(lambda self: None if self.__dict__["_metrics_world"] is None else self._m(True).rmse_loss)
    l1_loss_worldframe = property(lambda self: None if self.__dict__["_metrics_world"] is None else self._m(True).l1_loss)

    def body_frame_to_world_frame(self, batch_r_quat, grf_bodyFrame):
        return self._m().body_frame_to_world_frame(batch_r_quat, grf_bodyFrame)

    def calculate_losses_step_original(self, y: torch.Tensor, y_pred: torch.Tensor):
        Base_Lightning.calculate_losses_step(self, y, y_pred)

    def calculate_losses_step_worldframe(self, y, y_pred, batch_r_quat, test_only_on_z: bool = False):
        self.calculate_losses_step_original(y, y_pred)
        y_world = self.body_frame_to_world_frame(batch_r_quat, y.detach())
        y_pred_world = self.body_frame_to_world_frame(batch_r_quat, y_pred.detach())
        if test_only_on_z:
            y_world, y_pred_world = y_world[:, [2, 5, 8, 11]], y_pred_world[:, [2, 5, 8, 11]]
        self._m(True).calculate_losses_step(y_world, y_pred_world)

    def calculate_losses_step(self, y, y_pred, batch_r_quat=None):
        if self.body_to_world_frame:
            if batch_r_quat is None:
                raise ValueError("grf_body_to_world_frame needs the batch's r_o quaternions")
            self.calculate_losses_step_worldframe(y, y_pred, batch_r_quat)
        else:
            self.calculate_losses_step_original(y, y_pred)

    def log_losses_worldframe(self, step_name: str, on_step: bool):
        self.log_losses(step_name, on_step)
        for label, attr in _REG:
            self.log(f"{step_name}_{label}_WorldFrame", getattr(self, attr + "_worldframe"), on_step=on_step, on_epoch=not on_step)

    def calculate_losses_epoch_worldframe(self) -> None:
        self.calculate_losses_epoch()
        self._m(True).calculate_losses_epoch()

    def reset_all_metrics_worldframe(self) -> None:
        self.reset_all_metrics()
        self._m(True).reset_all_metrics()

    def _step(self, batch):
        y, y_pred = self.step_helper_function(batch)
        if self.body_to_world_frame:
            self.calculate_losses_step_worldframe(y, y_pred, batch.r_o.view(batch.batch_size, 4))
        else:
            self.calculate_losses_step_original(y, y_pred)

    def training_step(self, batch, batch_idx):
        r = self._fused_step(batch)
        if r is None:
            self._step(batch)
        else:
            y, y_pred, loss = r
            if self.body_to_world_frame:
                self.calculate_losses_step_worldframe(y, y_pred, batch.r_o.view(batch.batch_size, 4))
            else:
                self.calculate_losses_step_original(y, y_pred)
            self._m().set_step_loss(loss)
        (self.log_losses_worldframe if self.body_to_world_frame else self.log_losses)("train", on_step=True)
        return self._loss()

    def validation_step(self, batch, batch_idx):
        self._step(batch)
        return self._loss()

    def on_validation_epoch_start(self):
        (self.reset_all_metrics_worldframe if self.body_to_world_frame else self.reset_all_metrics)()

    def on_validation_epoch_end(self):
        if self.body_to_world_frame:
            self.calculate_losses_epoch_worldframe()
            self.log_losses_worldframe("val", on_step=False)
        else:
            self.calculate_losses_epoch()
            self.log_losses("val", on_step=False)

    def on_test_epoch_start(self):
        self.on_validation_epoch_start()

    def on_test_epoch_end(self):
        if self.body_to_world_frame:
            self.calculate_losses_epoch_worldframe()
            self.log_losses_worldframe("test", on_step=False)
        else:
            self.calculate_losses_epoch()
            self.log_losses("test", on_step=False)


# ---- centroidal-momentum wrappers (src/ms_hgnn/lightning_py/gnnLightning_com.py) ---------------------------------------------------
_COM_LOGS = (("MSE_loss", "mse_loss"), ("RMSE_loss", "rmse_loss"), ("MSE_loss_lin", "mse_loss_lin"), ("MSE_loss_ang", "mse_loss_ang"),
             ("cos_sim_lin", "cos_sim_lin"), ("cos_sim_ang", "cos_sim_ang"), ("avg_cos_sim", "avg_cos_sim"), ("loss", "loss"))


class COM_Base_Lightning(_Base):
    """`COM_Base_Lightning` (gnnLightning_com.py:28-232).  `data_path` is the dataset folder whose `processed/rss_stats.npz` holds the
    label statistics (`y_mean`, `y_std`) the cosine-similarity metrics un-standardise with (:52-58); `stats=(y_mean, y_std)` hands them
    over directly (synthetic data, tests)."""

    def __init__(self, optimizer: str, lr: float, data_path=None, stats=None):
        super().__init__()
        self.optimizer, self.lr, self.data_path = optimizer, lr, data_path
        self.regression = True
        if stats is None:
            import os
            import numpy as np
            if data_path is None:
                raise ValueError("COM wrappers need data_path (processed/rss_stats.npz) or stats=(y_mean, y_std)")
            st = np.load(os.path.join(str(data_path), "processed", "rss_stats.npz"))
            stats = (st["y_mean"], st["y_std"])
        self.__dict__["_stats"] = stats
        self.__dict__["_metrics"] = None
        self.logged = {}

    def _m(self):
        if self.__dict__["_metrics"] is None:
            from .metrics import ComStepMetrics
            self.__dict__["_metrics"] = ComStepMetrics(self.model.num_bases, *self.__dict__["_stats"])
        return self.__dict__["_metrics"]

    if _L is None:
        def log(self, name, value, on_step: bool = False, on_epoch: bool = True, **_):
            self.logged[name] = value

    def log_losses(self, step_name: str, on_step: bool):
        for label, attr in _COM_LOGS:
            self.log(f"{step_name}_{label}", getattr(self, attr), on_step=on_step, on_epoch=not on_step)

    def calculate_losses_step(self, y: torch.Tensor, y_pred: torch.Tensor):
        self._m().calculate_losses_step(y, y_pred)

    def calculate_losses_epoch(self) -> None:
        self._m().calculate_losses_epoch()

    def reset_all_metrics(self) -> None:
        self._m().reset_all_metrics()

    fused_training_step = Base_Lightning.fused_training_step

    def training_step(self, batch, batch_idx):
        r = _model_fused_step(self, batch)
        if r is None:
            y, y_pred = self.step_helper_function(batch)
            self.calculate_losses_step(y, y_pred)
        else:
            y, y_pred = self._shape(batch, r[0])
            self.calculate_losses_step(y, y_pred)
            self._m().set_step_loss(r[1])
        self.log_losses("train", on_step=True)
        return self.loss

    def on_validation_epoch_start(self):
        self.reset_all_metrics()

    def validation_step(self, batch, batch_idx):
        y, y_pred = self.step_helper_function(batch)
        self.calculate_losses_step(y, y_pred)
        return self.loss

    def on_validation_epoch_end(self):
        self.calculate_losses_epoch()
        self.log_losses("val", on_step=False)

    def on_test_epoch_start(self):
        self.reset_all_metrics()

    def test_step(self, batch, batch_idx):
        return self.validation_step(batch, batch_idx)

    def on_test_epoch_end(self):
        self.calculate_losses_epoch()
        self.log_losses("test", on_step=False)

    def configure_optimizers(self):
        return Base_Lightning.configure_optimizers(self)

    def step_helper_function(self, batch):
        """Outputs and labels per window, [batch, num_bases * 6] (gnnLightning_com.py:324-340, 394-409)."""
        return self._shape(batch, self.model(x_dict=batch.x_dict, edge_index_dict=batch.edge_index_dict))

    def _shape(self, batch, out_raw):
        batch_size = batch.batch_size if hasattr(batch, "batch_size") else 1
        width = self.model.num_bases * self.model.num_dimensions_per_base
        return torch.reshape(batch.y, (batch_size, width)), torch.reshape(out_raw.squeeze(), (batch_size, width))

    def _finish_init(self, dummy_batch):
        with torch.no_grad():
            self.model(x_dict=dummy_batch.x_dict, edge_index_dict=dummy_batch.edge_index_dict)
        if _L is not None:  # pragma: no cover
            self.save_hyperparameters(ignore=["dummy_batch", "activation_fn"])


for _n in ("mse_loss", "rmse_loss", "mse_loss_lin", "mse_loss_ang", "cos_sim_lin", "cos_sim_ang", "avg_cos_sim", "loss"):
    setattr(COM_Base_Lightning, _n, property(lambda self, _n=_n: None if self.__dict__.get("_metrics") is None else getattr(self.__dict__["_metrics"], _n)))


class COM_HGNN_Lightning(COM_Base_Lightning):
    """`COM_HGNN` (one base node, no symmetry; gnnLightning_com.py:290-340)."""

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), com_dimension: int = 6, data_path=None, stats=None):
        super().__init__(optimizer, lr, data_path, stats)
        self.model = models.COM_HGNN(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata, regression=regression,
                                     activation_fn=activation_fn, com_dimension=com_dimension)
        self.regression = regression
        self._finish_init(dummy_batch)


class COM_HGNN_SYM_Lightning(COM_Base_Lightning):
    """`COM_HGNN_K4` / `COM_HGNN_C2` / `COM_HGNN_S4` by `model_type` (gnnLightning_com.py:343-409)."""

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, dummy_batch, optimizer: str = "adam", lr: float = 0.003,
                 regression: bool = True, activation_fn=nn.ReLU(), symmetry_mode: Optional[str] = None, group_operator_path: Optional[str] = None,
                 model_type: str = "heterogeneous_gnn_k4_com", data_path=None, stats=None):
        super().__init__(optimizer, lr, data_path, stats)
        common = dict(hidden_channels=hidden_channels, num_layers=num_layers, data_metadata=data_metadata, regression=regression,
                      activation_fn=activation_fn)
        if model_type == "heterogeneous_gnn_k4_com":
            self.model = models.COM_HGNN_K4(symmetry_mode=symmetry_mode, group_operator_path=group_operator_path, **common)
        elif model_type == "heterogeneous_gnn_c2_com":
            self.model = models.COM_HGNN_C2(symmetry_mode=symmetry_mode, group_operator_path=group_operator_path, **common)
        elif model_type == "heterogeneous_gnn_s4_com":
            self.model = models.COM_HGNN_S4(**common)
        else:
            raise ValueError(f"unknown model_type '{model_type}'")
        self.regression = regression
        self._finish_init(dummy_batch)


class GraphedTrainingStep:
    """One training step of a wrapper -- `optimizer.zero_grad(); loss = wrapper.training_step(batch, i); loss.backward(); optimizer.step()` -- captured ONCE in a
    HIP graph and replayed per batch.

    Why: at the reference's own batch size (32, train_regression-grf_msgn.py:93) a step is a handful of short launches and the Python between them (autograd,
    metric bookkeeping, the optimizer) costs more than the GPU work: 0.32 ms eager against 0.13 ms replayed on one MI355X.  From ~4 000 windows on the step is
    GPU-bound and the replay buys nothing.

    The graph reads its inputs from STATIC device tensors (copies of `example_batch`'s tensors, made here); `__call__(batch)` copies the new batch's tensors
    into them (same shapes and dtypes) and replays.  Everything the step touches lives on the device: the loss / metric sums (metrics.py), the flat gradient
    buffer, and the optimizer's step count -- which is why the optimizer must be `FlatAdam(graph_safe=True)` (set `wrapper.graph_safe_optimizer = True`
    before `configure_optimizers()`) or a `torch.optim` optimizer created with `capturable=True`.  The model's parameters, the optimizer state and the metric
    state are restored after the warm-up steps the capture needs, so constructing this object does not train.

    Returns the step's loss as a device tensor (a static buffer: read it before the next call)."""

    def __init__(self, wrapper, optimizer, example_batch, warmup: int = 3):
        import copy
        from .optim import FlatAdam
        if isinstance(optimizer, FlatAdam) and not optimizer._graph_safe:
            raise ValueError("GraphedTrainingStep needs FlatAdam(graph_safe=True): set wrapper.graph_safe_optimizer = True before configure_optimizers()")
        if not isinstance(optimizer, FlatAdam) and not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainingStep needs an optimizer whose step count lives on the device (capturable=True)")
        self.wrapper, self.optimizer = wrapper, optimizer
        self.batch = self._static_copy(example_batch)
        dev = next(wrapper.parameters()).device
        snap_p = [p.detach().clone() for p in wrapper.parameters()]
        flat_opt = isinstance(optimizer, FlatAdam)
        if flat_opt:      # FlatAdam's state lives in flat device buffers the captured launches will address: snapshot / restore them IN PLACE
            optimizer._sync_steps()
            snap_o = None if optimizer._m is None or optimizer._owner is None else (optimizer._m.clone(), optimizer._v.clone(), int(optimizer._t))
        else:
            snap_o = copy.deepcopy(optimizer.state_dict())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager_step()
        with torch.no_grad():      # un-train: parameters and optimizer state as they were (the metric sums of the warm-up steps are cleared with the epoch's)
            for p, q in zip(wrapper.parameters(), snap_p):
                p.copy_(q)
            if flat_opt:
                if optimizer._m is None or optimizer._t_dev is None:
                    raise RuntimeError("GraphedTrainingStep: the optimizer did not take the flat route during capture (parameters or gradients are not the flat views)")
                if snap_o is None:
                    optimizer._m.zero_(); optimizer._v.zero_(); optimizer._t = 0
                else:
                    optimizer._m.copy_(snap_o[0]); optimizer._v.copy_(snap_o[1]); optimizer._t = snap_o[2]
                optimizer._t_dev.fill_(optimizer._t)
                optimizer._sync_steps()
            else:
                optimizer.load_state_dict(snap_o)
        if hasattr(wrapper, "reset_all_metrics"):
            wrapper.reset_all_metrics()

    @staticmethod
    def _static_copy(batch):
        import types
        out = types.SimpleNamespace()
        for k, v in vars(batch).items():
            if torch.is_tensor(v):
                setattr(out, k, v.detach().clone())
            elif isinstance(v, dict) and k == "x_dict":
                setattr(out, k, {kk: vv.detach().clone() for kk, vv in v.items()})
            else:
                setattr(out, k, v)      # edge_index_dict, batch_size, ...: the same for every batch of this size
        return out

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.wrapper.training_step(self.batch, 0)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def load(self, batch):
        """Copy a batch's tensors into the static ones (host or device sources; same shapes)."""
        for k, v in vars(batch).items():
            dst = getattr(self.batch, k, None)
            if torch.is_tensor(v) and torch.is_tensor(dst):
                dst.copy_(v, non_blocking=True)
            elif isinstance(v, dict) and k == "x_dict":
                for kk, vv in v.items():
                    dst[kk].copy_(vv, non_blocking=True)

    def __call__(self, batch=None):
        if batch is not None:
            self.load(batch)
        self.graph.replay()
        return self.loss
