"""A1-C2 at hidden = 256 (generic-width engine), 8192 windows: per-kernel times of the one-call step under a job-kernel mode (MSHGNN_GEN_TILE)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from morphsym_hgnn_amd import engine as eng, synth
spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 256, 3)
B = 8192
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
params = synth.make_params(3, spec.param_shapes())
if len(sys.argv) > 1: os.environ["MSHGNN_GEN_TILE"] = sys.argv[1]
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, params, e.device)
for _ in range(3): out, loss, g = e.step_mse(xs, flat, yd, B)
e.profile(True)
for _ in range(10): out, loss, g = e.step_mse(xs, flat, yd, B)
torch.cuda.synchronize()
st = {r["name"]: round(1e3 * r["total_ms"] / r["launches"], 1) for r in e.profile_read() if r["launches"]}
print("mode", os.environ.get("MSHGNN_GEN_TILE", "default"), "loss", float(loss), "sum us", round(sum(st.values()), 1), st)
