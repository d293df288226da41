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
