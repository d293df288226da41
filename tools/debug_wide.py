import sys, torch
sys.path.insert(0, ".")
from tests import helpers
from morphsym_hgnn_amd import engine as eng
name, dtype = sys.argv[1], sys.argv[2]
case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
B = case["B"]
e = eng.Engine(spec, dtype)
flat = eng.flatten_params(spec, params, e.device)
xs_cast = e.cast_inputs(x_dict)
out_a = e.forward(xs_cast, flat, B).clone()
X0a = e.hidden_state(B, 0).clone()
xw = e.cast_inputs({k: v.to(e.device) for k, v in x_dict.items()})
out_b = e.forward(xw, flat, B).clone()
X0b = e.hidden_state(B, 0).clone()
torch.cuda.synchronize()
print("out diff", float((out_a - out_b).abs().max()))
_, need = spec.node_liveness()
print("need", need[0])
for t, a, b in zip(e.types, xs_cast, list(xw)):
    n, F = spec.num_nodes[t], spec.widths[t]
    d = (a.view(B, n, -1)[:, :, :F].float() - b.view(B, n, -1)[:, :, :F].float()).abs()
    print(t, "rows diff per node", d.amax(dim=(0, 2)).tolist())
    bad = (d > 0).nonzero()
    print("  first bad", bad[:8].tolist(), "count", bad.shape[0])
d0 = (X0a.float() - X0b.float()).abs()
print("X0 diff per node", d0.amax(dim=(0, 2)).tolist() if d0.dim() == 3 else d0.max())
bad = (d0 > 0).nonzero()
print("X0 bad windows", sorted(set(bad[:, 0].tolist()))[:40], "nodes", sorted(set(bad[:, 1].tolist())))
t = "joint"; i = e.types.index(t)
n, F = spec.num_nodes[t], spec.widths[t]
a = xs_cast[i].view(B, n, -1)[:, :, :F]; b = list(xw)[i].view(B, n, -1)[:, :, :F]
bad = (a != b).nonzero()
bad = bad[(bad[:, 1] == 1)]
xj = x_dict[t].view(B, n, F)
for w_, n_, k_ in bad[:4].tolist():
    d = xj[w_, n_, k_]
    print("elem", w_, n_, k_, d.item().hex(), "cpu->bf16", d.to(torch.bfloat16).item(), "cpu f32", d.float().item().hex(), "f32->bf16 cpu", d.float().to(torch.bfloat16).item(),
          "gpu f64->bf16", d.cuda().to(torch.bfloat16).item(), "gpu f32->bf16", d.float().cuda().to(torch.bfloat16).item(), "cast path", a[w_, n_, k_].item(), "kernel", b[w_, n_, k_].item())
