"""Shared helpers for the parity tests: rebuild a golden case (spec, inputs, weights) from its fixture."""
import json
import os

import numpy as np
import torch
import yaml

from morphsym_hgnn_amd import synth, topology
from morphsym_hgnn_amd.spec import ModelSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg")

GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_group(name):
    if not name:
        return None, None
    path = os.path.join(CFG, name + ".yaml")
    with open(path) as f:
        return yaml.safe_load(f), path


def make_spec(kind, topo_name, cfg_name, hidden, layers, regression=True, grf=3):
    topo = topology.TOPOLOGIES[topo_name]()
    group, _ = load_group(cfg_name)
    widths = synth.feature_widths(kind, regression)
    return ModelSpec(kind=kind, topology=topo, hidden=hidden, num_layers=layers, widths=widths,
                     regression=regression, grf_dimension=grf, group=group)


def load_case(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = json.loads(str(fx["config"]))
    spec = make_spec(case["kind"], case["topo"], case["cfg"], case["hidden"], case["layers"],
                     case["regression"], case["grf"])
    seed = int(fx["seed"])
    B = case["B"]
    x_dict, y = synth.make_windows(seed, B, spec.num_nodes, spec.widths,
                                   spec.out_channels * 4 if case["regression"] else 4,
                                   classification=not case["regression"])
    params = synth.make_params(seed, spec.param_shapes())
    ei = spec.topology.edge_index_dict(B)
    return case, spec, fx, x_dict, y, params, ei


def sample_indices(name, numel, k=64):
    u = synth.det_uniform(7, "idx:" + name, (k,), 0.0, 1.0).numpy()
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def oracle_config(spec):
    from oracle import ms_hgnn_oracle as orc
    return orc.OracleConfig(kind=spec.kind, num_layers=spec.num_layers, edge_types=spec.edge_types,
                            regression=spec.regression, grf_dimension=spec.grf_dimension, group=spec.group,
                            num_timesteps=spec.num_timesteps)


def check_against_fixture(fx, out, loss, grads, rtol, what=""):
    """Compare (out, loss, grads) with a golden fixture.  Tolerance is relative to each tensor's max-abs
    (outputs) / L2 norm (gradients)."""
    ref_out = torch.from_numpy(fx["out"])
    scale = float(ref_out.abs().max())
    err = float((out.double().cpu().reshape(ref_out.shape) - ref_out).abs().max()) / scale
    assert err <= rtol, f"{what} output rel err {err:.3e} > {rtol}"
    if loss is not None:
        lerr = abs(float(loss) - float(fx["loss"])) / abs(float(fx["loss"]))
        assert lerr <= rtol, f"{what} loss rel err {lerr:.3e} > {rtol}"
    worst = (0.0, None)
    for key in fx.files:
        if not key.startswith("gnorm:"):
            continue
        name = key[len("gnorm:"):]
        g = grads[name].double().cpu().flatten().numpy()
        gn = float(fx[key])
        idx = sample_indices(name, g.size)
        denom = max(gn / np.sqrt(g.size), 1e-30)  # rms magnitude of the reference gradient
        if gn == 0.0:
            assert np.abs(g).max() == 0.0, f"{what} grad {name} should be exactly zero"
            continue
        e1 = float(np.abs(g[idx] - fx["gsample:" + name]).max()) / denom
        e2 = abs(float(np.sqrt((g ** 2).sum())) - gn) / gn
        e = max(e1 / 30.0, e2)  # samples: allow 30x the rms-relative tolerance (outliers of tiny entries)
        if e > worst[0]:
            worst = (e, name)
        assert e2 <= rtol, f"{what} grad-norm {name} rel err {e2:.3e} > {rtol}"
        assert e1 <= 30 * rtol, f"{what} grad samples {name} err/rms {e1:.3e} > {30 * rtol}"
    return err, worst
