"""Generic engine debugging aid: hidden states of a small h = 512 model under one job-kernel mode (MSHGNN_GEN_TILE), saved to gpurun_out/dbg_gen_<mode>.pt;
`debug_gen.py cmp` compares modes 3 and 4 (first differing layer, node, windows and columns)."""
import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from morphsym_hgnn_amd import engine as eng, synth, topology
from morphsym_hgnn_amd.spec import ModelSpec
if sys.argv[1] == "cmp":
    a = torch.load("gpurun_out/dbg_gen_3.pt"); b = torch.load(f"gpurun_out/dbg_gen_{sys.argv[2] if len(sys.argv) > 2 else 4}.pt")
    print("out equal", torch.equal(a["out"], b["out"]), float((a["out"] - b["out"]).abs().max()))
    for l, (x, z) in enumerate(zip(a["hs"], b["hs"])):
        d = (x - z).abs()
        print("X", l, "equal", torch.equal(x, z), "max", float(d.max()))
        if d.max() > 0 and d.dim() == 3:
            print("   per node:", [round(float(v), 3) for v in d.amax(dim=(0, 2))])
    sys.exit(0)
B = 256; L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(2), hidden=512, num_layers=L, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
params = synth.make_params(3, spec.param_shapes())
os.environ["MSHGNN_GEN_TILE"] = sys.argv[1]
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); flat = eng.flatten_params(spec, params, e.device)
out = e.forward(xs, flat, B, training=True).clone()
torch.cuda.synchronize()
hs = [e.hidden_state(B, l).float().cpu() for l in range(L + 1)]
torch.save({"out": out.cpu(), "hs": hs}, f"gpurun_out/dbg_gen_{sys.argv[1]}.pt")
print("mode", sys.argv[1], "out", out.flatten()[:6].tolist())
