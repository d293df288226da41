"""Host-time profile of the nn.Module surface (cProfile over N steps, GPU work asynchronous)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from morphsym_hgnn_amd import models, synth
from morphsym_hgnn_amd.checkpoint import load_into
torch.set_default_dtype(torch.float64)
spec = bench.build_spec(3)
dev = torch.device("cuda:0")
B = 8192
m = models.GRF_HGNN_C2(128, 3, spec.topology.metadata(), symmetry_mode="MorphSym", group_operator_path=os.path.join(bench.ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml"))
load_into(m, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
m.set_precision("bf16").to(dev)
x, y = bench.make_batch(spec, B, 99)
x64 = {k: v.to(dev, torch.float64) for k, v in x.items()}
y = y.to(dev, torch.float64).view(B, -1)
ei = spec.topology.edge_index_dict(B, device=dev)
with torch.no_grad():
    m(dict(x64), ei)
e = next(iter(m._engines.values()))
xin = dict(zip(e.types, e.cast_inputs(x64)))
def step():
    m.zero_grad(set_to_none=True)
    out = m(dict(xin), ei)
    loss = ((out.flatten() - y.flatten()) ** 2).mean()
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
