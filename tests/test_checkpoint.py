"""Lightning-checkpoint plumbing: weights saved in the wrapper's layout load by name."""
import torch

from morphsym_hgnn_amd import checkpoint as ck, models
from tests import helpers


def test_lightning_layout_round_trip(tmp_path):
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    _, cfg = helpers.load_group(case["cfg"])
    meta = spec.topology.metadata()
    # what the reference's ModelCheckpoint writes: model tensors under "model.", metric states next to them, constructor args
    ckpt = {"state_dict": {**{"model." + k: v for k, v in params.items()}, "metric_mse.sum_squared_error": torch.tensor(0.0),
                           "metric_mse.total": torch.tensor(0)},
            "hyper_parameters": {"hidden_channels": 128, "num_layers": 3, "data_metadata": meta, "regression": True, "optimizer": "adam",
                                 "lr": 1e-4, "symmetry_mode": "MorphSym", "group_operator_path": "/somewhere/else/a1-c2.yaml", "grf_dimension": 3}}
    path = tmp_path / "epoch=3-val_MSE_loss=1.0.ckpt"
    torch.save(ckpt, path)
    m = ck.model_from_checkpoint(str(path), "heterogeneous_gnn_c2", group_operator_path=cfg)     # override, as evaluate_model does
    assert isinstance(m, models.GRF_HGNN_C2)
    sd = m.state_dict()
    assert list(sd.keys()) == list(params.keys()) and all(torch.equal(sd[k], params[k]) for k in params)
    # an existing (still lazy) module, and the way back
    m2 = models.GRF_HGNN_C2(128, 3, meta, symmetry_mode="MorphSym", group_operator_path=cfg)
    ck.load_into(m2, ckpt)
    back = ck.to_lightning_checkpoint(m2, ckpt["hyper_parameters"])
    assert all(torch.equal(back["state_dict"]["model." + k], params[k]) for k in params)
    assert ck.model_state_dict(params).keys() == params.keys()       # bare state_dicts pass through


def test_checkpoint_with_foreign_objects_loads_without_importing_them(tmp_path):
    """The reference's save_hyperparameters() stores `dummy_batch` (a torch_geometric object) and `activation_fn` in the
    checkpoint; reading it must neither need those modules nor run their pickle code."""
    import sys
    import types
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    _, cfg = helpers.load_group(case["cfg"])
    mod = types.ModuleType("torch_geometric_standin_hetero_data")

    class HeteroDataBatch:      # what torch.save pickles by reference to its module
        def __init__(self):
            self.payload = {"x": torch.ones(3)}
    HeteroDataBatch.__module__ = mod.__name__
    HeteroDataBatch.__qualname__ = "HeteroDataBatch"
    mod.HeteroDataBatch = HeteroDataBatch
    sys.modules[mod.__name__] = mod
    try:
        ckpt = {"state_dict": {"model." + k: v for k, v in params.items()},
                "hyper_parameters": {"hidden_channels": 128, "num_layers": 3, "data_metadata": spec.topology.metadata(), "regression": True,
                                     "activation_fn": torch.nn.ReLU(), "dummy_batch": HeteroDataBatch(), "symmetry_mode": "MorphSym",
                                     "group_operator_path": cfg, "grf_dimension": 3}}
        path = tmp_path / "with_dummy_batch.ckpt"
        torch.save(ckpt, path)
    finally:
        del sys.modules[mod.__name__]
    loaded = ck.read_checkpoint(str(path))
    assert isinstance(loaded["hyper_parameters"]["dummy_batch"], ck.OpaqueObject)
    assert isinstance(loaded["hyper_parameters"]["activation_fn"], torch.nn.ReLU)
    m = ck.model_from_checkpoint(str(path), "heterogeneous_gnn_c2")
    sd = m.state_dict()
    assert all(torch.equal(sd[k], params[k]) for k in params)


def test_models_deepcopy_and_pickle_after_the_plan_exists():
    """copy.deepcopy / pickle / torch.save(model) work after the first forward compiled a plan (the reference modules support all
    three; Lightning's ddp_spawn pickles the module): engine handles and device scratch are dropped and rebuilt lazily."""
    import copy
    import ctypes
    import pickle
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case("a1c2_h128_L3_d3_B3")
    _, cfg = helpers.load_group(case["cfg"])
    m = models.GRF_HGNN_C2(128, 3, spec.topology.metadata(), symmetry_mode="MorphSym", group_operator_path=cfg)
    ck.load_into(m, {"state_dict": {"model." + k: v for k, v in params.items()}})
    m._spec = spec                                           # what the first forward leaves behind ...

    class FakeEngine:                                        # ... next to an engine holding ctypes handles
        def __init__(self):
            self.plan = ctypes.c_void_p(1234)
            self.lib = ctypes.pointer(ctypes.c_int(0))
    m._engines[("f32", "cuda:0")] = FakeEngine()
    m._checked_batches.add(3)
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone._engines == {} and clone._flat is None and clone._checked_batches == set()
        assert clone._spec.flat_size() == spec.flat_size()
        sd = clone.state_dict()
        assert all(torch.equal(sd[k], params[k]) for k in params)
    assert len(m._engines) == 1      # the original keeps its plan


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("name,model_type", [("a1c2_h128_L3_d3_B3", "heterogeneous_gnn_c2"), ("mck4_cls_h128_L2_B3", "heterogeneous_gnn_k4"),
                                             ("mi_h128_L2_d3_B2", "heterogeneous_gnn")])
def test_checkpoint_to_engine_matches_golden(name, model_type, tmp_path):
    """A Lightning-layout .ckpt (written the way the reference's ModelCheckpoint does, incl. a foreign `dummy_batch`) -> model_from_checkpoint
    -> forward + loss + backward on the GPU engine == the golden vectors of the reference run with those weights."""
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    _, cfg = helpers.load_group(case["cfg"])
    B = case["B"]
    hp = {"hidden_channels": 128, "num_layers": case["layers"], "data_metadata": spec.topology.metadata(), "regression": case["regression"],
          "activation_fn": torch.nn.ReLU(), "dummy_batch": ck.OpaqueObject(), "symmetry_mode": "MorphSym" if cfg else None,
          "group_operator_path": cfg, "grf_dimension": case["grf"]}
    path = tmp_path / "epoch=9.ckpt"
    torch.save({"state_dict": {**{"model." + k: v for k, v in params.items()}, "metric_mse.total": torch.tensor(0)}, "hyper_parameters": hp}, path)
    m = ck.model_from_checkpoint(str(path), model_type).cuda()
    xd = {k: v.cuda() for k, v in x_dict.items()}
    eid = {k: v.cuda() for k, v in ei.items()}
    out = m(x_dict=xd, edge_index_dict=eid)
    w = m.out_channels_per_foot * 4
    y_pred = torch.reshape(out.squeeze(), (B, w))
    if case["regression"]:
        loss = ((y_pred.flatten() - y.cuda().reshape(B, w).flatten()) ** 2).mean()
    else:
        loss = torch.nn.functional.cross_entropy(y_pred.reshape(B * 4, 2), y.cuda().reshape(B, 4).long().flatten())
    loss.backward()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()}
    helpers.check_against_fixture(fx, out.detach().cpu(), loss.detach().cpu(), grads, rtol=1e-4, what=name)


def test_checkpoint_reader_never_calls_functions_of_allowed_packages(tmp_path):
    """A crafted .ckpt whose pickle REDUCEs callables that live inside otherwise harmless packages (torch.utils.*, numpy.testing.*, os, builtins.eval)
    must load with those calls replaced by inert placeholders: the reader resolves exact (module, name) pairs, not package prefixes."""
    import os

    marker = tmp_path / "pwned"

    class Call:
        def __init__(self, fn, *args):
            self.fn, self.args = fn, args

        def __reduce__(self):
            return self.fn, self.args

    import numpy.testing._private.utils as npt
    import torch.utils.collect_env as ce
    cmd = f"echo PWNED > {marker}"
    ckpt = {"state_dict": {"model.w": torch.ones(3)},
            "hyper_parameters": {"a": Call(ce.run, cmd), "b": Call(os.system, cmd), "c": Call(npt.runstring, f"open({str(marker)!r}, 'w').write('x')", {}),
                                 "d": Call(eval, f"open({str(marker)!r}, 'w').write('x')"), "e": Call(torch.load, str(marker)), "lr": 1e-4}}
    path = tmp_path / "crafted.ckpt"
    torch.save(ckpt, path)
    loaded = ck.read_checkpoint(str(path))
    assert not marker.exists()
    assert all(isinstance(loaded["hyper_parameters"][k], ck.OpaqueObject) for k in "abcde")
    assert loaded["hyper_parameters"]["lr"] == 1e-4 and torch.equal(loaded["state_dict"]["model.w"], torch.ones(3))
