"""One-call step time of the bf16 plan at small batches (the reference trains with batch_size 30-64): wall time per step and per-kernel times."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
spec = bench.build_spec(3)
e = eng.Engine(spec, "bf16")
for B in (32, 64, 256, 1024, 2048):
    x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, 12)
    xs = e.cast_inputs(x_dict); yd = y.reshape(-1).to(e.device, torch.float32); flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    for _ in range(20): e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    e.profile(True)
    for _ in range(20): e.step_mse(xs, flat, yd, B)
    torch.cuda.synchronize()
    st = {r["name"]: round(1e3 * r["total_ms"] / r["launches"], 1) for r in e.profile_read() if r["launches"]}
    e.profile(False)
    print(f"B={B:5d}: {dt * 1e6:7.1f} us/step = {B / dt / 1e3:8.1f} k windows/s   kernels {st}")
