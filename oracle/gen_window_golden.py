"""ORACLE TOOLING -- runs ONLY in the build container (needs /root/reference).  Pins oracle/window_oracle.py against the
reference's own window-building functions and writes tests/golden/windows_a1c2.npz.

The reference's dataset module is imported BY FILE (oracle/import_stubs provides import-only stand-ins for rosbags /
torchvision / urchin, oracle/pyg_restated the HeteroData container) and its methods
`QuadSDKDataset_A1.load_data_at_dataset_seq[_3d]`, `QuadSDKDataset_NewGraph.load_data_sorted_c2` and
`.get_helper_heterogeneous_gnn_c2` are run on a stub `self` that carries a deterministic synthetic sequence (there is no
dataset in this image).  For every case the oracle must equal the reference exactly (max abs diff 0; 1e-12 relative for body-frame labels, where
scipy builds the rotation matrix with a different operation order)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "pyg_restated"))
sys.path.insert(0, os.path.join(HERE, "import_stubs"))
REF = "/root/reference/src/ms_hgnn"

from oracle import window_oracle as wo  # noqa: E402

JOINT_PERM = np.array([6, 7, 8, 0, 1, 2, 9, 10, 11, 3, 4, 5], dtype=np.uint)     # a non-identity order (quadSDKDataset.py:247-258)
FOOT_PERM = np.array([2, 0, 3, 1], dtype=np.uint)


def synthetic_sequence(seed, N):
    """float32-representable values so that fixtures and device inputs are exact."""
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.integers(-2000, 2000, size=s).astype(np.float64) / 64.0
    seq = {"imu_acc": f(N, 3), "imu_omega": f(N, 3), "q": f(N, 12), "qd": f(N, 12), "tau": f(N, 12), "F": f(N, 12) * 4,
           "r_p": f(N, 3), "r_o": f(N, 4), "timestamps": f(N, 3)}
    seq["r_o"][:, 3] += 40.0        # keep the quaternions away from zero norm
    return seq


def minicheetah_sequence(seed, N):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.integers(-2000, 2000, size=s).astype(np.float64) / 64.0
    return {"imu_acc": f(N, 3), "imu_omega": f(N, 3), "q": f(N, 12), "qd": f(N, 12), "tau_est": f(N, 12), "p": f(N, 12), "v": f(N, 12),
            "contacts": rng.integers(0, 2, size=(N, 4)).astype(np.float64)}


def solo_sequence(seed, N):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.integers(-2000, 2000, size=s).astype(np.float64) / 64.0
    return {"X": f(N, 24), "Y": f(N, 6)}


SOLO_KINDS = {"k4_com": ("heterogeneous_gnn_k4_com", 4, "load_data_sorted_k4"), "c2_com": ("heterogeneous_gnn_c2_com", 2, "load_data_sorted_c2"),
              "s4_com": ("heterogeneous_gnn_s4_com", 1, "load_data_sorted")}


def reference_module():
    pk = types.ModuleType("ms_hgnn"); pk.__path__ = [REF]; sys.modules["ms_hgnn"] = pk
    dp = types.ModuleType("ms_hgnn.datasets_py"); dp.__path__ = [REF + "/datasets_py"]; sys.modules["ms_hgnn.datasets_py"] = dp
    return importlib.import_module("ms_hgnn.datasets_py.quadSDKDataset_Morph")


def stub_dataset(mod, seq, T, grf_dimension, body_frame, normalize):
    s = types.SimpleNamespace()
    s.mat_data = seq; s.history_length = T; s.grf_dimension = grf_dimension; s.grf_body_to_world_frame = body_frame
    s.normalize = normalize; s.symmetry_operator = None
    s.joint_node_indices_sorted = JOINT_PERM; s.foot_node_indices_sorted = FOOT_PERM
    s.hgnn_number_nodes = (2, 12, 4); s.base_width = 6 * T; s.joint_width = 3 * T; s.foot_width = 1
    s.variables_to_use_base = np.array([0, 1]); s.variables_to_use_joint = np.array([0, 1, 2]); s.variables_to_use_foot = np.array([])
    s.urdf_name_to_graph_index_joint = {str(i): i for i in range(12)}; s.urdf_name_to_graph_index_foot = {f"f{i}": i for i in range(4)}
    z = torch.zeros(2, 0, dtype=torch.long)
    for k in ("bj_front", "jb_front", "bj_back", "jb_back", "jj", "fj", "jf", "bb"):
        setattr(s, k, z)
    A1, NG = mod.QuadSDKDataset_A1, mod.QuadSDKDataset_NewGraph
    s.load_data_at_dataset_seq_3d = types.MethodType(A1.load_data_at_dataset_seq_3d, s)
    s.load_data_at_dataset_seq = types.MethodType(A1.load_data_at_dataset_seq, s)
    s.load_data_sorted_c2 = types.MethodType(NG.load_data_sorted_c2, s)
    s.get = types.MethodType(NG.get_helper_heterogeneous_gnn_c2, s)
    return s


CASES = [dict(name="d3", grf=3, body=False, norm=False), dict(name="d3_body", grf=3, body=True, norm=False),
         dict(name="d1", grf=1, body=False, norm=False), dict(name="d3_norm", grf=3, body=False, norm=True)]


def main():
    mod = reference_module()
    T, N, seed = 150, 400, 20240915
    seq = synthetic_sequence(seed, N)
    starts = [0, 1, 37, 249, 250]
    fx = {"seed": np.array(seed), "N": np.array(N), "T": np.array(T), "starts": np.array(starts),
          "joint_perm": JOINT_PERM.astype(np.int64), "foot_perm": FOOT_PERM.astype(np.int64)}
    for c in CASES:
        ds = stub_dataset(mod, seq, T, c["grf"], c["body"], c["norm"])
        for st in starts:
            if c["norm"]:
                # The reference's standardisation line (quadSDKDataset_Morph.py:170) hands a torch tensor to np.nan_to_num(..., copy=False): numpy 1.x (what the
                # reference ran on) turned it into an array through Tensor.__array__ without a copy and replaced the NaNs in place; numpy 2.x (this image) refuses
                # `copy=False` on an object it has to convert.  The REFERENCE'S OWN branch is executed here under a one-function numpy-1.x shim -- np.nan_to_num
                # converting a non-array first, exactly what 1.x did -- in this generator only (build container; nothing of it ships), and checked against the same
                # expression written in torch (what rounds 1-5 pinned the option by).
                _orig_nan_to_num = np.nan_to_num
                def _nan_to_num_numpy1(x, copy=True, nan=0.0, posinf=None, neginf=None):
                    return _orig_nan_to_num(x if isinstance(x, np.ndarray) else np.asarray(x), copy=copy, nan=nan, posinf=posinf, neginf=neginf)
                np.nan_to_num = _nan_to_num_numpy1
                try:
                    ds.normalize = True
                    data = ds.get(st)
                finally:
                    np.nan_to_num = _orig_nan_to_num
                rb, rj, rf, ry = data["base"].x.numpy(), data["joint"].x.numpy(), data["foot"].x.numpy(), data.y.numpy()
                ds.normalize = False
                raw = ds.get(st)
                def nz(x, nvar, axes):   # rows [node][var][axis][T] -> standardise every T-run
                    v = x.reshape(x.shape[0], nvar * axes, T)
                    v = torch.nan_to_num((v - v.mean(dim=2, keepdim=True)) / v.std(dim=2, correction=1, keepdim=True), nan=0.0)
                    return v.reshape(x.shape[0], -1).numpy()
                for got, want, what in ((rb, nz(raw["base"].x, 2, 3), "base"), (rj, nz(raw["joint"].x, 3, 1), "joint")):
                    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), ("normalize=True: reference branch vs torch expression", what)
                assert np.array_equal(rf, raw["foot"].x.numpy()) and np.array_equal(ry, raw.y.numpy())
            else:
                data = ds.get(st)
                rb, rj, rf, ry = data["base"].x.numpy(), data["joint"].x.numpy(), data["foot"].x.numpy(), data.y.numpy()
            ob, oj, of, oy, oq = wo.a1_c2_window(seq, st, T, JOINT_PERM.astype(int), FOOT_PERM.astype(int), c["grf"], c["body"], c["norm"])
            for a, b, what in ((rb, ob, "base"), (rj, oj, "joint"), (rf, of, "foot"), (ry, oy, "y")):
                tol = 1e-12 * np.abs(a).max() if ((what == "y" and c["body"]) or c["norm"]) else 0.0   # rotated labels: scipy's matrix vs the closed form
                assert a.shape == b.shape and np.abs(a - b).max() <= tol, (c["name"], st, what, np.abs(a - b).max())
            if c["body"]:
                assert np.abs(data.r_o.numpy() - oq).max() == 0.0
            # fixture: the label vector and a strided sample of the features (the full rows are a pure function of the seed)
            fx[f"{c['name']}:{st}:y"] = ry
            fx[f"{c['name']}:{st}:base"] = rb[:, ::7].copy()
            fx[f"{c['name']}:{st}:joint"] = rj[:, ::11].copy()
        print(c["name"], "oracle == reference on", len(starts), "windows")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "windows_a1c2.npz"), **fx)

    # MiniCheetah K4 (LinTzuYaunDataset_Morph.py): same recipe, foot features and contact labels
    lm = importlib.import_module("ms_hgnn.datasets_py.LinTzuYaunDataset_Morph")
    seq4 = minicheetah_sequence(seed + 1, N)
    fx4 = {"seed": np.array(seed + 1), "N": np.array(N), "T": np.array(T), "starts": np.array(starts),
           "joint_perm": JOINT_PERM.astype(np.int64), "foot_perm": FOOT_PERM.astype(np.int64)}
    cls = next(getattr(lm, n) for n in dir(lm) if n.startswith("LinTzuYaunDataset") and hasattr(getattr(lm, n), "load_data_sorted_k4"))
    base_cls = importlib.import_module("ms_hgnn.datasets_py.LinTzuYaunDataset").LinTzuYaunDataset
    s4 = types.SimpleNamespace()
    s4.mat_data = seq4; s4.history_length = T; s4.normalize = False; s4.symmetry_operator = None; s4.swap_legs = None
    s4.joint_node_indices_sorted = JOINT_PERM; s4.foot_node_indices_sorted = FOOT_PERM
    s4.hgnn_number_nodes = (4, 12, 4); s4.base_width = 6 * T; s4.joint_width = 2 * T; s4.foot_width = 6 * T
    s4.variables_to_use_base = np.array([0, 1]); s4.variables_to_use_joint = np.array([0, 1]); s4.variables_to_use_foot = np.array([0, 1])
    s4.urdf_name_to_graph_index_joint = {str(i): i for i in range(12)}; s4.urdf_name_to_graph_index_foot = {f"f{i}": i for i in range(4)}
    z = torch.zeros(2, 0, dtype=torch.long)
    for k in ("bj", "jb", "jj", "fj", "jf", "gt", "gs", "bj_attr", "jb_attr", "jj_attr", "fj_attr", "jf_attr", "gt_attr", "gs_attr"):
        setattr(s4, k, z)
    s4.load_data_at_dataset_seq = types.MethodType(base_cls.load_data_at_dataset_seq, s4)
    s4.load_data_sorted_k4 = types.MethodType(cls.load_data_sorted_k4, s4)
    s4.get = types.MethodType(cls.get_helper_heterogeneous_gnn, s4)
    for st in starts:
        data = s4.get(st)
        ob, oj, of, oy = wo.minicheetah_k4_window(seq4, st, T, JOINT_PERM.astype(int), FOOT_PERM.astype(int))
        for a, b, what in ((data["base"].x.numpy(), ob, "base"), (data["joint"].x.numpy(), oj, "joint"), (data["foot"].x.numpy(), of, "foot"),
                           (data.y.numpy(), oy, "y")):
            assert a.shape == b.shape and np.abs(a - b).max() == 0.0, ("k4", st, what)
        fx4[f"k4:{st}:y"] = data.y.numpy(); fx4[f"k4:{st}:base"] = data["base"].x.numpy()[:, ::7].copy()
        fx4[f"k4:{st}:joint"] = data["joint"].x.numpy()[:, ::11].copy(); fx4[f"k4:{st}:foot"] = data["foot"].x.numpy()[:, ::13].copy()
    print("k4 oracle == reference on", len(starts), "windows")
    # normalize=True -- what BASELINE configs[2] trains with (train_classification_msgn.py passes it): the reference's own branch (LinTzuYaunDataset_Morph.py:337-345)
    # under the numpy-1.x nan_to_num shim described above
    _orig_nan_to_num = np.nan_to_num
    np.nan_to_num = lambda x, copy=True, nan=0.0, posinf=None, neginf=None: _orig_nan_to_num(x if isinstance(x, np.ndarray) else np.asarray(x), copy=copy, nan=nan, posinf=posinf, neginf=neginf)
    try:
        s4.normalize = True
        for st in starts:
            data = s4.get(st)
            ob, oj, of, oy = wo.minicheetah_k4_window(seq4, st, T, JOINT_PERM.astype(int), FOOT_PERM.astype(int), normalize=True)
            for a, b, what in ((data["base"].x.numpy(), ob, "base"), (data["joint"].x.numpy(), oj, "joint"), (data["foot"].x.numpy(), of, "foot"), (data.y.numpy(), oy, "y")):
                assert a.shape == b.shape and np.abs(a - b).max() <= 1e-12 * max(np.abs(a).max(), 1.0), ("k4_norm", st, what, np.abs(a - b).max())
            fx4[f"k4_norm:{st}:y"] = data.y.numpy(); fx4[f"k4_norm:{st}:base"] = data["base"].x.numpy()[:, ::7].copy()
            fx4[f"k4_norm:{st}:joint"] = data["joint"].x.numpy()[:, ::11].copy(); fx4[f"k4_norm:{st}:foot"] = data["foot"].x.numpy()[:, ::13].copy()
    finally:
        np.nan_to_num = _orig_nan_to_num
        s4.normalize = False
    print("k4 normalize=True: oracle == reference branch on", len(starts), "windows")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "windows_mck4.npz"), **fx4)

    # Solo-12 centroidal-momentum task (soloDataset.py), K4 / C2 / S4 graphs, history 1 (the COM models' setting) and 5
    sm = importlib.import_module("ms_hgnn.datasets_py.soloDataset")
    scls = next(getattr(sm, n) for n in dir(sm) if isinstance(getattr(sm, n), type) and hasattr(getattr(sm, n), "load_data_sorted_k4"))
    flex = importlib.import_module("ms_hgnn.datasets_py.flexibleDataset").FlexibleDataset
    sq = solo_sequence(seed + 2, N)
    fxs = {"seed": np.array(seed + 2), "N": np.array(N), "starts": np.array(starts), "joint_perm": JOINT_PERM.astype(np.int64)}
    for kind, (model_type, nb, loader) in SOLO_KINDS.items():
        for Th in (1, 5):
            ss = types.SimpleNamespace()
            ss.X, ss.Y, ss.history_length, ss.model_type = sq["X"], sq["Y"], Th, model_type
            ss.symmetry_operator = None; ss.swap_legs = None; ss.normalize = False
            ss.joint_node_indices_sorted = JOINT_PERM; ss.foot_node_indices_sorted = FOOT_PERM
            ss.hgnn_number_nodes = (nb, 12, 0); ss.base_width = 6 * Th; ss.joint_width = 2 * Th
            ss.urdf_name_to_graph_index_joint = {str(i): i for i in range(12)}
            z = torch.zeros(2, 0, dtype=torch.long)
            for k in ("bj", "jb", "jj", "gt", "gs", "bj_front", "jb_front", "bj_back", "jb_back", "bb"):
                setattr(ss, k, z)
            ss.load_data_at_dataset_seq = types.MethodType(scls.load_data_at_dataset_seq, ss)
            ss.load_data_sorted = types.MethodType(scls.load_data_sorted, ss)
            ss.load_data_sorted_k4 = types.MethodType(scls.load_data_sorted_k4, ss)
            ss.load_data_sorted_c2 = types.MethodType(scls.load_data_sorted_c2, ss)
            ss.find_variables_to_use = types.MethodType(flex.find_variables_to_use, ss)
            ss.get = types.MethodType(scls.get_helper_heterogeneous_gnn, ss)
            for st in starts:
                data = ss.get(st)
                ob, oj, oy = wo.solo_com_window(sq["X"], sq["Y"], st, Th, JOINT_PERM.astype(int), nb)
                for a, b, what in ((data["base"].x.numpy(), ob, "base"), (data["joint"].x.numpy(), oj, "joint"), (data.y.numpy(), oy, "y")):
                    assert a.shape == b.shape and np.abs(a - b).max() == 0.0, (kind, Th, st, what)
                fxs[f"{kind}:{Th}:{st}:y"] = data.y.numpy(); fxs[f"{kind}:{Th}:{st}:joint"] = data["joint"].x.numpy()
        print(kind, "oracle == reference on", 2 * len(starts), "windows")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "windows_solo.npz"), **fxs)


if __name__ == "__main__":
    main()
