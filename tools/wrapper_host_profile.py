"""cProfile of the eager wrapper step (zero_grad + training_step + backward + FlatAdam.step) at the reference's batch size: where the host time goes.
usage: python tools/wrapper_host_profile.py [B]"""
import cProfile, io, os, pstats, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import synth, wrappers
from morphsym_hgnn_amd.checkpoint import load_into
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spec = bench.build_spec(3)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")
x, y = bench.make_batch(spec, B, 5)
torch.set_default_dtype(torch.float64)
x64 = {k: v.to(dev, torch.float64) for k, v in x.items()}
ei = spec.topology.edge_index_dict(B, device=dev)
os.environ["MSHGNN_DTYPE"] = "bf16"
w = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, spec.num_layers, spec.topology.metadata(), types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei),
                                   lr=1e-4, symmetry_mode="MorphSym", group_operator_path=cfg)
load_into(w.model, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
w.model.set_precision("bf16"); w.to(dev)
opt = w.configure_optimizers()
batch = types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei, y=y.to(dev, torch.float64).view(B, -1), batch_size=B)


def wstep():
    opt.zero_grad(set_to_none=True)
    l = w.training_step(batch, 0)
    l.backward()
    opt.step()


for _ in range(20): wstep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): wstep()
torch.cuda.synchronize(); print(f"B={B}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per eager wrapper step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): wstep()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:7000])
