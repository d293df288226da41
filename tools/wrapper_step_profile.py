"""The wrapper surface's GPU work per step at the headline batch, fp64 device inputs (rocprofv3 --kernel-trace --stats -- python3 tools/wrapper_step_profile.py):
zero_grad + training_step + backward + FlatAdam.step, 60 steps."""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import synth, wrappers
from morphsym_hgnn_amd.checkpoint import load_into
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
spec = bench.build_spec(3)
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")
torch.set_default_dtype(torch.float64)
x, y = bench.make_batch(spec, B, 5)
x64 = {k: v.to(dev, torch.float64) for k, v in x.items()}
ei = spec.topology.edge_index_dict(B, device=dev)
os.environ["MSHGNN_DTYPE"] = "bf16"
w = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, spec.num_layers, spec.topology.metadata(), types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei),
                                   lr=1e-4, symmetry_mode="MorphSym", group_operator_path=cfg)
load_into(w.model, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
w.model.set_precision("bf16"); w.to(dev)
opt = w.configure_optimizers()
batch = types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei, y=y.to(dev, torch.float64).view(B, -1), batch_size=B)
for _ in range(60):
    opt.zero_grad(set_to_none=True)
    l = w.training_step(batch, 0)
    l.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(l))
