"""Where the nn.Module surface's step goes: host enqueue time vs GPU time (bench.py's module_surface workload, plan-dtype inputs).
   python tools/module_host_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from morphsym_hgnn_amd import models, synth
from morphsym_hgnn_amd.checkpoint import load_into

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
device_loss = len(sys.argv) > 2 and sys.argv[2] == "device-loss"      # loss + metrics from metrics.StepMetrics instead of torch's elementwise kernels
wrapper = len(sys.argv) > 2 and sys.argv[2] == "wrapper"              # wrappers.HGNN_C2_Lightning_Reg.training_step (one-call engine step) + backward
dev = torch.device("cuda", 0)
B = 8192
spec = bench.build_spec(3, "a1c2", 128)
torch.set_default_dtype(torch.float64)
cfg = os.path.join(bench.ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")
m = models.GRF_HGNN_C2(128, 3, spec.topology.metadata(), symmetry_mode="MorphSym", group_operator_path=cfg)
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


from morphsym_hgnn_amd.metrics import StepMetrics
sm = StepMetrics(regression=True, device=dev)
y32 = y.float()


if wrapper:
    import types
    from morphsym_hgnn_amd import wrappers
    w = wrappers.Base_Lightning.__new__(wrappers.HGNN_C2_Lightning_Reg)
    wrappers.Base_Lightning.__init__(w, "adam", 1e-4, True)
    w.model = m


def step():
    m.zero_grad(set_to_none=True)
    if wrapper:
        w.training_step(types.SimpleNamespace(x_dict=dict(xin), edge_index_dict=ei, y=y32, batch_size=B), 0).backward()
        return
    out = m(dict(xin), ei)
    if device_loss:
        sm.calculate_losses_step(y32, out)
        loss = sm.mse_loss
    else:
        loss = ((out.flatten() - y.flatten()) ** 2).mean()
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
# (a) wall per step, GPU kept busy
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
# (b) host enqueue time alone: sync before each step, time only the python part
host = 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host += time.perf_counter() - t0
host /= steps
# (c) GPU time of a step in isolation (events around it, queue empty)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for a, b in ev:
    torch.cuda.synchronize()
    a.record(); step(); b.record()
torch.cuda.synchronize()
gpu = sorted(a.elapsed_time(b) for a, b in ev)[steps // 2]
print(f"wall {wall*1e3:.3f} ms/step   host enqueue {host*1e3:.3f} ms/step   GPU span of an isolated step (median) {gpu:.3f} ms")
# split of the host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
