"""Generic-width engine: the tile modes of its job kernel (MSHGNN_GEN_TILE: 0 = 4 waves, 1 = 8 waves on 128-window tiles, 2 = 8 waves, 3 = 16 waves on 64-window
tiles, 6 = k_gstep4, 8 = k_gstep5: the default) on the synthetic 32-limb model (h = 512, L = 6): per-kernel times of the one-call
step; bf16 modes give identical bits.  `check_gen_modes.py 1024 6`; DT=x3 for the split arithmetic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec = bench.build_spec_config("synth32") if hasattr(bench, "build_spec_config") else None
if spec is None:
    from morphsym_hgnn_amd import topology
    from morphsym_hgnn_amd.spec import ModelSpec
    spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
params = synth.make_params(3, spec.param_shapes())
res = {}
for mode in sys.argv[2:] or ["3"]:
    os.environ["MSHGNN_GEN_TILE"] = mode
    e = eng.Engine(spec, os.environ.get("DT", "bf16"))
    xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, params, e.device)
    for _ in range(2): out, loss, g = e.step_mse(xs, flat, yd, B)
    e.profile(True)
    for _ in range(5): out, loss, g = e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize()
    st = {r["name"]: round(1e3 * r["total_ms"] / r["launches"], 1) for r in e.profile_read() if r["launches"]}
    e.profile(False)
    res[mode] = (out.clone(), loss.clone(), g.clone())
    print("mode", mode, "loss", float(loss), "sum us", round(sum(st.values()), 1), st)
ks = list(res)
for k in ks[1:]:
    print("mode", k, "vs", ks[0], ": out equal", torch.equal(res[k][0], res[ks[0]][0]), " grads equal", torch.equal(res[k][2], res[ks[0]][2]),
          " max |dg|", float((res[k][2] - res[ks[0]][2]).abs().max()))
