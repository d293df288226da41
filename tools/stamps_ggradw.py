"""Phase clocks of k_ggradw's step loop (build with EXTRA=-DGGW_STAMPS into another library and pass it as MSHGNN_LIB): per workgroup, summed over its steps,
wave 0's clocks [at the step's barrier, waiting for the next step's loads, staging it, (second barrier: the two-barrier loop of the split plan only),
requests + MFMAs]."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphsym_hgnn_amd import engine as eng, synth, topology
from morphsym_hgnn_amd.spec import ModelSpec
B = 1024
spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
os.environ["MSHGNN_GGW_STAMPS"] = hex(stamps.data_ptr())
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
for _ in range(2): e.step_mse(xs, flat, yd, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64); s = s[s[:, 5] > 0]
for name, sel in (("lean super-units", (s[:, 6].astype(int) & 1) == 1), ("general super-units", (s[:, 6].astype(int) & 1) == 0)):
    q = s[sel]
    if not len(q): continue
    per = q[:, :5] / q[:, 5:6]
    print(f"{name}: {len(q)} workgroups, steps per workgroup median {np.median(q[:, 5]):.0f}; clocks per step (median over workgroups):")
    for k, nm in enumerate(["barrier", "wait for loads", "stage", "(barrier 2)", "requests + MFMAs"]):
        print(f"  {nm:18s} {np.median(per[:, k]):8.1f}   p90 {np.percentile(per[:, k], 90):8.1f}")
    print(f"  {'sum':18s} {np.median(per.sum(1)):8.1f}")
