"""Lightning checkpoint compatibility.

The reference trains through Lightning (`gnnLightning.py:1346-1380`, `ModelCheckpoint`) and evaluates with
`<Wrapper>.load_from_checkpoint(path, ...)` (`gnnLightning.py:955-990`).  A Lightning `.ckpt` is a `torch.save`d dict whose
`"state_dict"` holds the wrapper's tensors -- the model's under the prefix `model.` (`self.model = GRF_HGNN_C2(...)`,
`gnnLightning.py:580-590`) next to metric states -- and whose `"hyper_parameters"` holds the wrapper's constructor
arguments (`save_hyperparameters()`, `:596`).  The drop-in modules keep the reference's parameter names, so published
weights load by name; this module does the prefix / lazy-encoder / hyper-parameter plumbing without Lightning.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

import builtins
import io
import pickle

from . import models

MODEL_CLASSES = {
    # model_type strings of train_model / evaluate_model (gnnLightning.py:957-990, 1290-1330)
    "heterogeneous_gnn": models.GRF_HGNN, "heterogeneous_gnn_k4": models.GRF_HGNN_K4, "heterogeneous_gnn_c2": models.GRF_HGNN_C2,
    "heterogeneous_gnn_k4_com": models.COM_HGNN_K4, "heterogeneous_gnn_c2_com": models.COM_HGNN_C2,
    "heterogeneous_gnn_s4_com": models.COM_HGNN_S4,
}


# ---- reading a Lightning .ckpt without Lightning / torch_geometric ---------------------------------------------------
# The reference wrappers call save_hyperparameters() with no ignore list (gnnLightning.py:444,494,543,596), so
# ckpt["hyper_parameters"] holds `dummy_batch` (a torch_geometric HeteroData batch) and `activation_fn`.  A plain
# torch.load(weights_only=False) then needs torch_geometric importable (it is not part of this stack) and runs arbitrary
# pickle code of a third-party file.  Only `state_dict` and the scalar hyper-parameters are needed here, so the file is
# read with weights_only=True first and, when that reader rejects a global, with an unpickler that resolves ONLY the exact
# (module, name) pairs below -- the constructors a tensor / ndarray / plain-container pickle needs.  Every other global
# (torch_geometric.*, lightning.*, but also torch.utils.*, torch.hub.*, numpy.testing.* ...: a package prefix is not a
# licence, those hold functions that run commands) becomes an inert placeholder whose REDUCE / BUILD do nothing.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch.nn.parameter", "Parameter"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.storage", "TypedStorage"), ("torch.storage", "UntypedStorage"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("_codecs", "encode"), ("copyreg", "_reconstructor"),
    ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"), ("pathlib", "Path"), ("argparse", "Namespace"),
}
# activation modules a wrapper may have stored as `activation_fn` (parameter-free nn.Modules, rebuilt through copyreg._reconstructor)
_SAFE_ACTIVATIONS = {"ReLU", "LeakyReLU", "ELU", "GELU", "SiLU", "Tanh", "Sigmoid", "Softplus", "PReLU", "SELU", "CELU", "Mish", "Hardtanh", "ReLU6",
                     "Identity"}
_SAFE_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                  "slice", "range", "object"}


def _safe_torch_attr(name):
    """`torch.<name>` for the value kinds a tensor pickle references by name: dtypes and (legacy) storage classes."""
    obj = getattr(torch, name, None)
    if isinstance(obj, torch.dtype):
        return obj
    if isinstance(obj, type) and issubclass(obj, (torch.storage.TypedStorage, torch.storage.UntypedStorage)):
        return obj
    return None


class OpaqueObject:
    """Stand-in for a pickled object whose class is not on the allow-list (e.g. the wrapper's `dummy_batch`)."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):      # the state of a foreign object is dropped, never interpreted
        pass

    def __call__(self, *args, **kwargs):
        return OpaqueObject()

    def __repr__(self):
        return "<OpaqueObject (class not importable / not allow-listed when the checkpoint was read)>"


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "builtins":
            return getattr(builtins, name) if name in _SAFE_BUILTINS else OpaqueObject
        if module == "torch" and "." not in name:
            obj = _safe_torch_attr(name)
            if obj is not None:
                return obj
        if (module, name) in _SAFE_GLOBALS or (module in ("torch.nn.modules.activation", "torch.nn.modules.linear") and name in _SAFE_ACTIVATIONS):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return OpaqueObject
        return OpaqueObject


class _restricted_pickle:
    """`pickle_module` for torch.load: torch subclasses `Unpickler` (storage handling) and calls up into find_class above."""
    __name__ = "morphsym_hgnn_amd.checkpoint._restricted_pickle"
    Unpickler = _RestrictedUnpickler
    load = staticmethod(lambda f, **kw: _RestrictedUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _RestrictedUnpickler(io.BytesIO(b), **kw).load())


def read_checkpoint(path) -> dict:
    """Load a Lightning `.ckpt` (or a bare state_dict file) onto the host without importing or calling what it pickled."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:      # hyper_parameters hold non-tensor objects the weights_only reader rejects: restricted reader
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_restricted_pickle)


def model_state_dict(ckpt: dict, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """The model's tensors of a Lightning checkpoint (or of a bare state_dict), prefix stripped, wrapper metric states dropped."""
    sd = ckpt.get("state_dict", ckpt)
    out = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if not out:      # already a bare model state_dict
        out = {k: v for k, v in sd.items() if isinstance(v, torch.Tensor) and not k.startswith("metric_")}
    return out


def load_into(model: nn.Module, ckpt, strict: bool = True):
    """Load a checkpoint (path or dict) into one of the drop-in modules: the lazy encoder is materialised from the
    checkpoint's weight shapes first (the reference does this with a dummy forward, gnnLightning.py:593-595)."""
    if not isinstance(ckpt, dict):
        ckpt = read_checkpoint(ckpt)
    sd = model_state_dict(ckpt)
    for t, lin in model.encoder.lins.items():
        w = sd.get(f"encoder.lins.{t}.weight")
        if w is not None:
            lin.materialize(int(w.shape[1]))
            ref = model.decoder.weight
            lin.to(device=ref.device, dtype=ref.dtype)
    return model.load_state_dict(sd, strict=strict)


def model_from_checkpoint(ckpt, model_type: str, data_metadata=None, **overrides) -> nn.Module:
    """Rebuild the model of a checkpoint from its `hyper_parameters` (constructor arguments of the Lightning wrapper) and
    load its weights.  `overrides` take precedence, exactly like the keyword arguments of `load_from_checkpoint`
    (`symmetry_mode`, `group_operator_path`, `grf_dimension`, ... -- gnnLightning.py:962-990)."""
    if not isinstance(ckpt, dict):
        ckpt = read_checkpoint(ckpt)
    if model_type not in MODEL_CLASSES:
        raise ValueError(f"unknown model_type {model_type!r}")
    hp = dict(ckpt.get("hyper_parameters", {}))
    hp.update(overrides)
    meta = data_metadata if data_metadata is not None else hp.get("data_metadata")
    if meta is None:
        raise ValueError("data_metadata is neither in the checkpoint's hyper_parameters nor given")
    cls = MODEL_CLASSES[model_type]
    kw = {"regression": hp.get("regression", True)}
    if cls in (models.GRF_HGNN_C2, models.GRF_HGNN_K4, models.COM_HGNN_K4, models.COM_HGNN_C2, models.COM_HGNN_S4):
        kw["symmetry_mode"] = hp.get("symmetry_mode")
        kw["group_operator_path"] = hp.get("group_operator_path")
    if cls in (models.GRF_HGNN_C2, models.GRF_HGNN):
        kw["grf_dimension"] = hp.get("grf_dimension", 3 if cls is models.GRF_HGNN_C2 else 1)
    model = cls(int(hp["hidden_channels"]), int(hp["num_layers"]), meta, **kw)
    load_into(model, ckpt, strict=True)
    return model


def to_lightning_checkpoint(model: nn.Module, hyper_parameters: Optional[dict] = None, prefix: str = "model.") -> dict:
    """The inverse: a dict in Lightning's layout whose `state_dict` the reference's wrappers load by name.  The reference's
    constructors need `dummy_batch` (gnnLightning.py:564-596), which is not stored here: pass it to `load_from_checkpoint`
    as a keyword argument next to the path, as `evaluate_model` passes its other overrides."""
    return {"state_dict": {prefix + k: v.detach().cpu() for k, v in model.state_dict().items()},
            "hyper_parameters": dict(hyper_parameters or {}), "pytorch-lightning_version": "2.1.0", "epoch": 0, "global_step": 0}
