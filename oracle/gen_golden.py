"""ORACLE TOOLING -- runs ONLY in the build container (needs /root/reference).  Generates tests/golden/*.npz.

What it does, per case:
  1. imports the reference's own model file (hgnn_c2.py / hgnn_k4.py / hgnn.py) BY FILE PATH, with
     oracle/pyg_restated on sys.path in place of the absent torch_geometric==2.5.0;
  2. builds a deterministic synthetic minibatch + weights (morphsym_hgnn_amd.synth), loads the weights into
     the reference module, runs forward + wrapper reshape + MSE/CE + backward in float64;
  3. runs oracle/ms_hgnn_oracle.py on the same inputs and asserts agreement <= 1e-12 (relative to the
     tensor's max-abs) on the output, the loss and EVERY parameter gradient, plus the exact parameter count;
  4. writes a small fixture: case config, seed, full output, loss, and for each gradient its L2 norm, sum,
     four +-1 random projections and 64 sampled entries (indices are a deterministic function of the name).  Inputs and weights are not
     stored -- tests regenerate them bit-identically from the seed.

Nothing from /root/reference is copied: fixtures hold numbers only.
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/src/ms_hgnn/lightning_py"

from morphsym_hgnn_amd import synth, topology  # noqa: E402
from morphsym_hgnn_amd.spec import ModelSpec  # noqa: E402
from oracle import ms_hgnn_oracle as orc  # noqa: E402

CASES = [
    # name, kind, topology, cfg yaml, hidden, layers, B, regression, grf_dim
    dict(name="a1c2_h128_L3_d3_B3", kind="c2", topo="a1-c2", cfg="a1-c2", hidden=128, layers=3, B=3, regression=True, grf=3),
    dict(name="a1c2_h128_L1_d1_B2", kind="c2", topo="a1-c2", cfg="a1-c2", hidden=128, layers=1, B=2, regression=True, grf=1),
    dict(name="a1c2_h128_L2_d3_B37", kind="c2", topo="a1-c2", cfg="a1-c2", hidden=128, layers=2, B=37, regression=True, grf=3),
    dict(name="a1c2_nosym_h128_L2_d3_B2", kind="c2", topo="a1-c2", cfg=None, hidden=128, layers=2, B=2, regression=True, grf=3),
    dict(name="mcc2_cls_h128_L2_B3", kind="c2", topo="mini_cheetah-c2", cfg="mini_cheetah-c2", hidden=128, layers=2, B=3, regression=False, grf=3),
    dict(name="mck4_cls_h128_L2_B3", kind="k4", topo="mini_cheetah-k4", cfg="mini_cheetah-k4", hidden=128, layers=2, B=3, regression=False, grf=3),
    dict(name="mck4_reg_h128_L1_B2", kind="k4", topo="mini_cheetah-k4", cfg="mini_cheetah-k4", hidden=128, layers=1, B=2, regression=True, grf=3),
    dict(name="mi_h128_L2_d1_B3", kind="mi", topo="quadruped-mi", cfg=None, hidden=128, layers=2, B=3, regression=True, grf=1),
    dict(name="mi_h128_L2_d3_B2", kind="mi", topo="quadruped-mi", cfg=None, hidden=128, layers=2, B=2, regression=True, grf=3),
    # the paper's depth (train_regression-grf_msgn.py:94) and SURVEY.md 8(d) config 3 (MiniCheetah K4 classification, L=8)
    dict(name="a1c2_h128_L8_d3_B2", kind="c2", topo="a1-c2", cfg="a1-c2", hidden=128, layers=8, B=2, regression=True, grf=3),
    dict(name="mck4_cls_h128_L8_B2", kind="k4", topo="mini_cheetah-k4", cfg="mini_cheetah-k4", hidden=128, layers=8, B=2, regression=False, grf=3),
    # Solo centroidal-momentum variants (decoder on base nodes, T=1): hgnn_k4_com.py, hgnn_c2_com.py, hgnn_s4_com.py, hgnn.py:COM_HGNN
    dict(name="solok4com_h128_L3_B5", kind="k4_com", topo="solo-k4-com", cfg="solo-k4", hidden=128, layers=3, B=5, regression=True, grf=3),
    dict(name="solok4com_nosym_h128_L1_B2", kind="k4_com", topo="solo-k4-com", cfg=None, hidden=128, layers=1, B=2, regression=True, grf=3),
    dict(name="soloc2com_h128_L2_B4", kind="c2_com", topo="solo-c2-com", cfg="solo-c2", hidden=128, layers=2, B=4, regression=True, grf=3),
    dict(name="solos4com_h128_L2_B3", kind="s4_com", topo="solo-s4-com", cfg=None, hidden=128, layers=2, B=3, regression=True, grf=3),
    dict(name="com_hgnn_h128_L2_B3", kind="s4_com", topo="solo-s4-com", cfg=None, hidden=128, layers=2, B=3, regression=True, grf=3, ref="COM_HGNN"),
    # BASELINE.json configs[4] / SURVEY.md 8(d) config 5: synthetic 32-limb robot, MI-HGNN (hgnn.py:GRF_HGNN) at h=512, 6 layers; plus
    # other widths (the reference's --hidden_size flag, research/train_regression-grf_msgn.py:95) on the 8-limb and the A1-C2 graphs
    dict(name="synth32_mi_h512_L6_B2", kind="mi", topo="synth32-mi", cfg=None, hidden=512, layers=6, B=2, regression=True, grf=3),
    dict(name="synth8_mi_h256_L3_B3", kind="mi", topo="synth8-mi", cfg=None, hidden=256, layers=3, B=3, regression=True, grf=1),
    dict(name="a1c2_h256_L2_d3_B3", kind="c2", topo="a1-c2", cfg="a1-c2", hidden=256, layers=2, B=3, regression=True, grf=3),
]


def load_reference_module(fname: str):
    shim = os.path.join(HERE, "pyg_restated")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sample_indices(name: str, numel: int, k: int = 64) -> np.ndarray:
    u = synth.det_uniform(7, "idx:" + name, (k,), 0.0, 1.0).numpy()
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def projection_signs(name: str, numel: int, k: int = 4) -> np.ndarray:
    import zlib
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode())))
    return rng.integers(0, 2, size=(k, numel)).astype(np.float64) * 2.0 - 1.0


def build_case(case):
    topo = topology.TOPOLOGIES[case["topo"]]()
    group = None
    cfg_path = None
    if case["cfg"]:
        cfg_path = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg", case["cfg"] + ".yaml")
        group = yaml.safe_load(open(cfg_path))
    widths = synth.feature_widths(case["kind"], case["regression"])
    spec = ModelSpec(kind=case["kind"], topology=topo, hidden=case["hidden"], num_layers=case["layers"],
                     widths=widths, regression=case["regression"], grf_dimension=case["grf"], group=group)
    return topo, spec, cfg_path


def run_reference(case, topo, spec, cfg_path, params, x_dict, ei, y):
    torch.set_default_dtype(torch.float64)  # gnnLightning.py:1183
    meta = topo.metadata()
    with contextlib.redirect_stdout(io.StringIO()):
        if case["kind"] == "c2":
            m = load_reference_module("hgnn_c2.py").GRF_HGNN_C2(
                case["hidden"], case["layers"], meta, regression=case["regression"],
                symmetry_mode="MorphSym" if cfg_path else None, group_operator_path=cfg_path,
                grf_dimension=case["grf"])
        elif case["kind"] == "k4":
            m = load_reference_module("hgnn_k4.py").GRF_HGNN_K4(
                case["hidden"], case["layers"], meta, regression=case["regression"],
                symmetry_mode="MorphSym" if cfg_path else None, group_operator_path=cfg_path)
        elif case["kind"] in ("k4_com", "c2_com"):
            f, cls = {"k4_com": ("hgnn_k4_com.py", "COM_HGNN_K4"), "c2_com": ("hgnn_c2_com.py", "COM_HGNN_C2")}[case["kind"]]
            m = getattr(load_reference_module(f), cls)(
                case["hidden"], case["layers"], meta, regression=True,
                symmetry_mode="MorphSym" if cfg_path else None, group_operator_path=cfg_path)
        elif case["kind"] == "s4_com" and case.get("ref") == "COM_HGNN":
            m = load_reference_module("hgnn.py").COM_HGNN(case["hidden"], case["layers"], meta, regression=True, com_dimension=6)
        elif case["kind"] == "s4_com":
            m = load_reference_module("hgnn_s4_com.py").COM_HGNN_S4(case["hidden"], case["layers"], meta, regression=True)
        else:
            m = load_reference_module("hgnn.py").GRF_HGNN(
                case["hidden"], case["layers"], meta, regression=case["regression"], grf_dimension=case["grf"])
    with torch.no_grad():  # lazy init, gnnLightning.py:593-595
        m({k: v.clone() for k, v in x_dict.items()}, ei)
    sd = m.state_dict()
    assert list(sd.keys()) == list(params.keys()), (list(sd.keys())[:5], list(params.keys())[:5])
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == spec.num_params(), (n_params, spec.num_params())
    m.load_state_dict(params)
    m.zero_grad()
    out = m({k: v.clone() for k, v in x_dict.items()}, ei)
    B = case["B"]
    if case["kind"].endswith("_com"):      # gnnLightning_com.py:323-340 (reshape) and :96-97,121 (MSE over everything)
        w = m.num_bases * m.num_dimensions_per_base
        y_pred = torch.reshape(out.squeeze(), (B, w))
        loss = ((y_pred.flatten() - torch.reshape(y, (B, w)).flatten()) ** 2).mean()
        loss.backward()
        grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
        return out.detach(), loss.detach(), grads, n_params
    # gnnLightning.py:691 reshapes to (B, out_channels_per_foot * 4): the wrapper hard-codes a quadruped's 4 feet.  The MSE that follows is
    # taken over the FLATTENED tensors (:633-639), so for the many-limb synthetic robot the same expression is evaluated on all n_foot rows.
    w = m.out_channels_per_foot * topo.num_nodes.get("foot", 4)
    y_pred = torch.reshape(out.squeeze(), (B, w))
    yy = torch.reshape(y, (B, w if case["regression"] else 4))  # gnnLightning.py:694 / :512
    if case["regression"]:
        loss = ((y_pred.flatten() - yy.flatten()) ** 2).mean()
    else:
        loss = torch.nn.functional.cross_entropy(y_pred.reshape(B * 4, 2), yy.long().flatten())
    loss.backward()
    # parameters that cannot influence the foot output (e.g. last-layer relations into base/joint) keep
    # grad=None in the reference; the fixture records them as zeros
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    return out.detach(), loss.detach(), grads, n_params


def main():
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    spath = os.path.join(ROOT, "tests", "golden", "SUMMARY.json")
    only = sys.argv[1:]                     # optional: case-name substrings; other fixtures are left untouched
    summary = json.load(open(spath)) if (only and os.path.exists(spath)) else {}
    for case in CASES:
        if only and not any(o in case["name"] for o in only):
            continue
        topo, spec, cfg_path = build_case(case)
        seed = 1234 + len(case["name"])
        B = case["B"]
        x_dict, y = synth.make_windows(seed, B, topo.num_nodes, spec.widths,
                                       spec.out_channels * topo.num_nodes[spec.out_type] if case["regression"] else 4,
                                       classification=not case["regression"])
        params = synth.make_params(seed, spec.param_shapes())
        ei = topo.edge_index_dict(B)

        ref_out, ref_loss, ref_grads, n_params = run_reference(case, topo, spec, cfg_path, params, x_dict, ei, y)

        ocfg = orc.OracleConfig(kind=case["kind"], num_layers=case["layers"], edge_types=topo.edge_types,
                                regression=case["regression"], grf_dimension=case["grf"], group=spec.group)
        o_out, o_loss, o_grads = orc.step(ocfg, params, x_dict, ei, y, B)

        def rel(a, b):
            return float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))

        errs = {"out": rel(o_out, ref_out), "loss": rel(o_loss, ref_loss)}
        for k in ref_grads:
            errs["grad:" + k] = rel(o_grads[k], ref_grads[k])
        worst = max(errs.values())
        assert worst <= 1e-12, (case["name"], max(errs, key=errs.get), worst)

        fx = {"out": ref_out.numpy(), "loss": np.array(float(ref_loss)), "seed": np.array(seed),
              "n_params": np.array(n_params), "config": np.array(json.dumps(case))}
        for k, g in ref_grads.items():
            flat = g.flatten().numpy()
            idx = sample_indices(k, flat.size)
            fx["gnorm:" + k] = np.array(np.sqrt((flat ** 2).sum()))
            fx["gsum:" + k] = np.array(flat.sum())
            fx["gsample:" + k] = flat[idx]
            fx["gproj:" + k] = projection_signs(k, flat.size) @ flat      # four +-1 projections (tests/helpers.py draws the same signs)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", case["name"] + ".npz"), **fx)
        summary[case["name"]] = {"n_params": n_params, "oracle_vs_reference_max_rel_err": worst,
                                 "loss": float(ref_loss)}
        print(f"{case['name']}: params={n_params} loss={float(ref_loss):.6g} oracle-vs-reference max rel err={worst:.2e}")
    with open(spath, "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
