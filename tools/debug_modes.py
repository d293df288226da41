import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from morphsym_hgnn_amd import engine as eng, synth
kind, topo, cfg, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
spec = helpers.make_spec(kind, topo, cfg, 128, 3, grf=3 if kind == "c2" else 1)
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(5, B, spec.num_nodes, spec.widths, n_y)
params = synth.make_params(5, spec.param_shapes())
res = {}
modes = {"slab": ("2", "1"), "slab-2launch": ("2", "0"), "8wave-step": ("0", "1"), "8wave": ("0", "0")}
for mode, (slab, step) in modes.items():
    os.environ["MSHGNN_SLAB"] = slab; os.environ["MSHGNN_STEP_KERNEL"] = step
    e = eng.Engine(spec, "bf16")
    xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, params, e.device)
    out, loss, g = e.step_mse(xs, flat, yd, B); torch.cuda.synchronize()
    res[mode] = (out.clone(), loss.clone(), g.clone())
ref = res["8wave"]
for mode in res:
    o, l, g = res[mode]
    ga, gb = eng.unflatten(spec, g), eng.unflatten(spec, ref[2])
    worst = max(((float((ga[k] - gb[k]).abs().max()), k) for k in ga), key=lambda t: t[0])
    nbad = int((o != ref[0]).sum())
    print(mode, "out equal", torch.equal(o, ref[0]), "n diff", nbad, "max", float((o - ref[0]).abs().max()), "first bad rows", (o != ref[0]).nonzero()[:4].tolist(), "loss", float(l), "worst grad", worst)
torch.save({m: [t.cpu() for t in r] for m, r in res.items()}, os.environ.get("DBG_OUT", "/tmp/dbg_modes.pt"))
