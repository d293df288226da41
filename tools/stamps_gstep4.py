"""Phase clocks of k_gstep4's K-chunk loop (build with EXTRA=-DGGW_STAMPS into another library and pass it as MSHGNN_LIB): per workgroup, summed over its
chunks, wave 0's clocks [barrier in front of the staging, wait for the chunk's source rows, LDS writes + second barrier, wait for the weight fragment, MFMAs]
and the epilogue.  usage: stamps_gstep4.py [launch name, default layer_fwd0]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphsym_hgnn_amd import engine as eng, synth, topology
from morphsym_hgnn_amd.spec import ModelSpec
B = 1024
which = sys.argv[1] if len(sys.argv) > 1 else "layer_fwd0"
spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
os.environ["MSHGNN_GS4_STAMPS"] = f"{which}:{hex(stamps.data_ptr())}"
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
for _ in range(2): e.step_mse(xs, flat, yd, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64); s = s[s[:, 6] > 0]
print(f"{which}: {len(s)} workgroups")
for nt in sorted(set(s[:, 7].astype(int))):
    q = s[s[:, 7].astype(int) == nt]
    per = q[:, :5] / q[:, 6:7]
    print(f"jobs of {nt} terms: {len(q)} workgroups, {np.median(q[:, 6]):.0f} chunks; clocks per chunk (median over workgroups), epilogue {np.median(q[:, 5]):.0f}:")
    names = ["barrier", "wait for rows", "LDS + barrier", "wait for weights", "MFMAs"]
    if os.environ.get("MSHGNN_GEN_TILE") == "8": names = ["top of the chunk", "K steps (MFMAs, weight and row requests)", "stage next rows", "barrier", "(prologue / chunks)"]      # k_gstep5
    for k, nm in enumerate(names):
        print(f"  {nm:34s} {np.median(per[:, k]):8.1f}   p90 {np.percentile(per[:, k], 90):8.1f}")
    print(f"  {'sum':18s} {np.median(per.sum(1)):8.1f}")
