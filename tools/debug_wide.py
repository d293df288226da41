"""Compare the workspace stashes of the wide stack kernels against the 8-wave ones on one case (first mismatching buffer / node)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from morphsym_hgnn_amd import engine as eng, synth
kind, topo, cfg, B = (sys.argv[1:5] + ["k4", "mini_cheetah-k4", "mini_cheetah-k4", "777"][len(sys.argv) - 1:])[:4]
B = int(B)
spec = helpers.make_spec(kind, topo, cfg, 128, 3, grf=3 if kind == "c2" else 1)
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(5, B, spec.num_nodes, spec.widths, n_y)
params = synth.make_params(5, spec.param_shapes())
res = {}
for mode, (wide, slab2, slab) in {"wide": ("2", "0", "2"), "slab2": ("0", "2", "2"), "8wave": ("0", "0", "0")}.items():
    os.environ["MSHGNN_WIDE"] = wide; os.environ["MSHGNN_SLAB2"] = slab2; os.environ["MSHGNN_SLAB"] = slab
    e = eng.Engine(spec, "bf16")
    xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, params, e.device)
    e.workspace(B).zero_()
    out, loss, g = e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize()
    st = {"out": out.clone(), "g": g.clone()}
    for l in range(4):
        st[f"x{l}"] = e.hidden_state(B, l).clone()
        st[f"dx{l}"] = e.grad_hidden(B, l).clone()
    res[mode] = st
NN = sum(spec.num_nodes.values())
for mode in ("wide", "slab2"):
  print("==", mode)
  for k in res[mode]:
    a, b = res[mode][k].float(), res["8wave"][k].float()
    if torch.equal(a, b): print(k, "equal"); continue
    d = (a - b).abs()
    if k in ("out", "g"): print(k, "DIFF max", float(d.max()), "elements", int((d > 0).sum())); continue
    per_node = d.amax(dim=(0, 2))
    wins = (d.amax(dim=(1, 2)) > 0).nonzero().flatten()
    print(k, "DIFF per node:", [round(float(v), 4) for v in per_node], "windows:", wins[:8].tolist(), "...", int(wins.numel()))
