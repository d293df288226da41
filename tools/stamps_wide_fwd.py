"""In-kernel phase timing of k_wide_fwd (build with EXTRA=-DMSHGNN_FS_STAMPS): clock64 stamps by thread 0 of every workgroup (one per CU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3); B = 8192
stamps = torch.zeros(512 * 32, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS"] = hex(stamps.data_ptr())
e = eng.Engine(spec, "bf16", device=dev)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
y = torch.randn(B * 12, generator=g).to(dev)
infer = os.environ.get("STAMP_INFER") == "1"     # forward without the training stashes
for _ in range(3):
    if infer: e.forward(xs, flat, B, training=False)
    else: e.step_mse(xs, flat, y, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(512, 32).astype(np.float64); s = s[s[:, 30] > 0]
names = {0: "start", 1: "tile staged", 30: "end (decoder + MSE tail)"}; order = [0, 1]
for l in range(3):
    names.update({2 + 4 * l: f"L{l} MAC phase", 3 + 4 * l: f"L{l} barrier", 4 + 4 * l: f"L{l} relu epilogue", 5 + 4 * l: f"L{l} base MLP + barrier"})
    order += [2 + 4 * l, 3 + 4 * l, 4 + 4 * l, 5 + 4 * l]
order.append(30)
prev = 0
for k in order:
    d = np.median(s[:, k] - s[:, prev]) if k else 0
    print(f"  {names[k]:28s} +{d:9.0f} cycles (median)   since start {np.median(s[:, k] - s[:, 0]):9.0f}")
    prev = k
for l in range(3):
    if s[:, 20 + l].max() > 0: print(f"  L{l}: accumulator init {np.median(s[:, 20 + l] - s[:, 1 if l == 0 else 5 + 4 * (l - 1)]):8.0f}   MAC engine {np.median(s[:, 2 + 4 * l] - s[:, 20 + l]):8.0f} cycles")
print("start spread (cycles):", int(s[:, 0].max() - s[:, 0].min()), " span:", int(s[:, 30].max() - s[:, 0].min()))
