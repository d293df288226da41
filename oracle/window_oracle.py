"""ORACLE (test infrastructure only): numpy restatement of the reference's per-window feature building for the
QuadSDK / A1 C2 dataset -- src/ms_hgnn/datasets_py/quadSDKDataset_Morph.py:99-175 (load_data_sorted_c2: base tiling, joint /
label re-ordering, optional per-window standardisation), :304-369 (get_helper_heterogeneous_gnn_c2: axis-major
flatten('F') per variable, all-ones feet, y, r_o) and :444-489 (load_data_at_dataset_seq[_3d]: window slicing, label =
GRF of the last step, optional world->body rotation).  Pinned by oracle/gen_window_golden.py, which runs those very
reference functions (imported by file) on the same synthetic sequence and asserts equality."""
import numpy as np


def quat_matrix(q):
    """Rotation matrix of a scalar-last quaternion (what scipy's Rotation.from_quat(q).as_matrix() returns)."""
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _standardise(a):
    """flexibleDataset.py:390-396: (x - mean) / std over the window rows, unbiased, NaN -> 0."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nan_to_num((a - a.mean(axis=0)) / a.std(axis=0, ddof=1), nan=0.0)


def a1_c2_window(seq, start, T, joint_perm, foot_perm, grf_dimension=3, body_frame_labels=False, normalize=False, n_base=2):
    """seq: dict of [N, .] arrays (imu_acc, imu_omega, q, qd, tau, F, r_o).  Returns (base [n_base, 6T], joint [12, 3T],
    foot [4, 1], y [4 * grf_dimension], r_o [4])."""
    sl = slice(start, start + T)
    lin, ang = np.tile(seq["imu_acc"][sl], (1, n_base)), np.tile(seq["imu_omega"][sl], (1, n_base))
    jp, jv, jt = (seq[k][sl][:, joint_perm] for k in ("q", "qd", "tau"))
    grfs = np.array(seq["F"][sl][-1], dtype=np.float64)
    quat = seq["r_o"][sl][-1]
    if body_frame_labels:
        grfs = (quat_matrix(quat) @ grfs.reshape(4, 3).T).T.flatten()
    if grf_dimension == 1:
        labels = grfs[[2, 5, 8, 11]][foot_perm]
    else:
        labels = grfs[[int(i * 3 + k) for i in foot_perm for k in range(3)]]
    if normalize:
        lin, ang, jp, jv, jt = (_standardise(a) for a in (lin, ang, jp, jv, jt))
    base = np.stack([np.concatenate([v[:, 3 * i:3 * i + 3].flatten("F") for v in (lin, ang)]) for i in range(n_base)])
    joint = np.stack([np.concatenate([v[:, i] for v in (jp, jv, jt)]) for i in range(len(joint_perm))])
    return base, joint, np.ones((4, 1)), labels, np.asarray(quat, dtype=np.float64)


def minicheetah_k4_window(seq, start, T, joint_perm, foot_perm, normalize=False, n_base=4):
    """MiniCheetah (LinTzuYaun) K4 graph -- LinTzuYaunDataset.py:65-88 (window slicing, contact labels of the last step),
    LinTzuYaunDataset_Morph.py:251-347 (`load_data_sorted_k4`: IMU tiled to the 4 base nodes, joint / foot / label
    re-ordering, optional standardisation), :555-625 (`get_helper_heterogeneous_gnn`: flatten('F') per variable).
    seq: imu_acc, imu_omega [N,3]; q, qd [N,12]; p, v [N,12]; contacts [N,4].  Returns base [4, 6T], joint [12, 2T],
    foot [4, 6T], y [4]."""
    sl = slice(start, start + T)
    lin, ang = np.tile(seq["imu_acc"][sl], (1, n_base)), np.tile(seq["imu_omega"][sl], (1, n_base))
    jp, jv = (seq[k][sl][:, joint_perm] for k in ("q", "qd"))
    fidx = [int(i * 3 + k) for i in foot_perm for k in range(3)]
    fp, fv = seq["p"][sl][:, fidx], seq["v"][sl][:, fidx]
    labels = np.asarray(seq["contacts"][sl][-1])[foot_perm]
    if normalize:
        lin, ang, jp, jv, fp, fv = (_standardise(a) for a in (lin, ang, jp, jv, fp, fv))
    base = np.stack([np.concatenate([v[:, 3 * i:3 * i + 3].flatten("F") for v in (lin, ang)]) for i in range(n_base)])
    joint = np.stack([np.concatenate([v[:, i] for v in (jp, jv)]) for i in range(len(joint_perm))])
    foot = np.stack([np.concatenate([v[:, 3 * i:3 * i + 3].flatten("F") for v in (fp, fv)]) for i in range(len(foot_perm))])
    return base, joint, foot, labels.astype(np.float64)



def solo_com_window(X, Y, start, T, joint_perm, n_base):
    """Solo-12 centroidal-momentum task -- soloDataset.py:382-400 (`load_data_at_dataset_seq`: q / qd = the two halves of X, label = Y of
    the window's last row, base IMU = zeros), :332-380 / :546-632 / :634-717 (`load_data_sorted[_k4|_c2]`: base series and labels tiled
    to the n_base base nodes, joints re-ordered; labels re-ordered to [lin(3) | ang(3)] per base node) and :235-300
    (`get_helper_heterogeneous_gnn`: flatten('F') per variable).  X: [N, 24], Y: [N, 6].  Returns base [n_base, 6T] (zeros),
    joint [12, 2T], y [n_base * 6]."""
    sl = slice(start, start + T)
    jp, jv = X[sl, :12][:, joint_perm], X[sl, 12:][:, joint_perm]
    lab = np.concatenate([Y[sl, :3][-1], Y[sl, 3:][-1]])
    base = np.zeros((n_base, 6 * T))
    joint = np.stack([np.concatenate([v[:, i] for v in (jp, jv)]) for i in range(len(joint_perm))])
    y = np.concatenate([np.concatenate([lab[:3], lab[3:]]) for _ in range(n_base)])
    return base, joint, y
