"""Time mshgnn_assemble_windows (8192 A1 windows out of a 200 000-step sequence) with both gather kernels, and the end-to-end loop of
examples/train_flat.py's shape: assembly + step + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe
N, B, T = 200000, 8192, 150
rng = np.random.default_rng(0)
seq = {k: rng.standard_normal((N, c)).astype(np.float32) for k, c in (("imu_acc", 3), ("imu_omega", 3), ("q", 12), ("qd", 12), ("tau", 12), ("F", 12), ("r_o", 4))}
jp, fp = list(range(12)), list(range(4))
starts = torch.from_numpy(rng.integers(0, N - T, size=B)).cuda()
for dtype in ("bf16", "f32"):
    for fast in (True, False):
        st = SequenceStore(seq, quadsdk_a1_c2_recipe(jp, fp, T, 3), dtype=dtype, fast=fast)
        for _ in range(5): st.assemble(starts, reuse_buffers=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): st.assemble(starts, reuse_buffers=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"assemble {dtype} {'chunk-gather' if fast else 'run-gather'}: {dt * 1e3:.4f} ms  ({B / dt / 1e6:.1f} M windows/s)")

# end to end: assembly + step (+ the same fused into the encoder), bf16 plan, A1-C2 L=3
import bench
from morphsym_hgnn_amd import engine as eng, synth as _synth
spec = bench.build_spec(3)
e = eng.Engine(spec, "bf16")
flat = eng.flatten_params(spec, _synth.make_params(0, spec.param_shapes()), e.device)
st = SequenceStore(seq, quadsdk_a1_c2_recipe(jp, fp, T, 3), dtype="bf16")
out = torch.empty(B * 4, 3, device="cuda"); gfl = torch.empty_like(flat); loss = torch.empty(1, device="cuda")
def two():
    xs, y, _ = st.assemble(starts, reuse_buffers=True)
    e.step_mse(xs, flat, y.reshape(-1), B, out=out, grad_flat=gfl, loss=loss)
def fused():
    e.step_mse_series(st, starts, flat, out=out, grad_flat=gfl, loss=loss, materialize=True)
def fused2():
    e.step_mse_series(st, starts, flat, out=out, grad_flat=gfl, loss=loss, materialize=False)
for name, fn in (("assemble + step_mse", two), ("step_mse_series, windows materialised by the encoder", fused), ("step_mse_series, no materialised windows", fused2)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{name}: {dt * 1e3:.4f} ms/step  ({B / dt / 1e6:.1f} M windows/s)")
# the same with the batch's window starts SORTED (an SGD minibatch is a set: the caller may order it): neighbouring windows overlap, so the gathers
# of neighbouring rows walk the same cache lines
starts = torch.sort(starts).values
for name, fn in (("sorted starts: assemble + step_mse", two), ("sorted starts: step_mse_series, windows materialised", fused), ("sorted starts: step_mse_series, no materialised windows", fused2)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{name}: {dt * 1e3:.4f} ms/step  ({B / dt / 1e6:.1f} M windows/s)")
e.profile(True)
for _ in range(20): fused2()
torch.cuda.synchronize()
print({s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 1) for s in e.profile_read() if s["launches"]})
