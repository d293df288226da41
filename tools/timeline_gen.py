"""Workgroup timeline of one launch of the generic engine (build with EXTRA=-DGEN_TIMELINE into another library and pass it as MSHGNN_LIB):
python tools/timeline_gen.py <launch name: layer_fwd1 | layer_bwd2 | gradw | ...> -- start / end wall clock (100 MHz) and CU of every workgroup; prints the
duration distribution, how many workgroups are resident over time, and how long the launch's tail is."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphsym_hgnn_amd import engine as eng, synth, topology
from morphsym_hgnn_amd.spec import ModelSpec
which = sys.argv[1] if len(sys.argv) > 1 else "layer_fwd1"
B = 1024
spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
tl = torch.zeros(8192 * 4, dtype=torch.int64, device="cuda")
os.environ["MSHGNN_GEN_TL"] = f"{which}:{hex(tl.data_ptr())}"
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
for _ in range(3): e.step_mse(xs, flat, yd, B)
torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(-1, 4); t = t[t[:, 1] > 0]
st, en = (t[:, 0] - t[:, 0].min()) / 100.0, (t[:, 1] - t[:, 0].min()) / 100.0      # us
dur = en - st
cu = (t[:, 3] & 0xf) * 1000 + ((t[:, 2] >> 8) & 0xf) + 16 * ((t[:, 2] >> 13) & 0x7) + 128 * ((t[:, 2] >> 12) & 1)      # (xcc, se, cu) -> a key
print(f"{which}: {len(t)} workgroups on {len(np.unique(cu))} CUs, launch {en.max():.1f} us; workgroup duration min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
print(f"sum of durations / (CUs x launch) = {dur.sum() / (256 * en.max()):.3f}")
for q in np.linspace(0, en.max(), 11)[:-1]:
    print(f"  t = {q:7.1f} us: {int(((st <= q) & (en > q)).sum()):4d} workgroups resident")
order = np.argsort(st)
print("last 12 starts (us):", np.round(st[order][-12:], 1), " their durations:", np.round(dur[order][-12:], 1))
