"""In-kernel phase timing of k_gradw_bf16 (clock64 deltas accumulated by thread 0 of every workgroup over its steps).
Build the library with  make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_GW_STAMPS=1  first.
Measured (r01, B=8192): 5300 cycles per 64-window step = barrier 290 + wait loads / LDS write 1900 + barrier 220 + ISSUE of
the next step's 12 loads 2030 (memory-pipe back-pressure) + MFMA phase 910: the kernel is bound by bytes moved per CU."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3); B = 8192
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS_GW"] = hex(stamps.data_ptr())
e = eng.Engine(spec, "bf16", device=dev)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
y = torch.randn(B * 12, generator=g).to(dev)
for _ in range(3): e.step_mse(xs, flat, y, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 5] > 0]
steps = s[:, 5]
names = ["barrier (wait MFMAs of all waves)", "wait loads + LDS write", "barrier", "issue next loads", "MFMA phase (LDS tr reads + MFMAs)"]
print("workgroups", len(s), "steps/WG median", np.median(steps), "cycles/step median", np.median(s[:, :5].sum(1) / steps))
for k, nm in enumerate(names):
    print(f"  {nm:40s} {np.median(s[:, k] / steps):8.0f} cycles/step")
