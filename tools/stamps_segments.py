"""Per-segment clocks of k_stack_fwd's MAC phases (build with EXTRA=-DMSHGNN_SEG_STAMPS): how long each weight pack's
walk takes on every wave, i.e. whether the next pack's fragment arrives in time."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3); B = 8192
stamps = torch.zeros(512 * 32 + 512 * 384, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS"] = hex(stamps.data_ptr())
e = eng.Engine(spec, "bf16", device=dev)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
for _ in range(3): e.forward(xs, flat, B, training=True)
torch.cuda.synchronize()
s = stamps.cpu().numpy()
ph = s[:512 * 32].reshape(512, 32).astype(np.float64)
seg = s[512 * 32:].reshape(512, 3, 8, 16).astype(np.float64)
for l in range(3):
    print(f"layer {l}: MAC phase (wave 0 stamps) median {np.median(ph[:, 3 + 4 * l] - ph[:, 2 + 4 * l]):.0f} cycles")
    for wv in (0, 4, 1, 5):
        t = seg[:, l, wv, :]
        n = int((t[0] > 0).sum()) - 1
        d = np.median(t[:, 1:n + 1] - t[:, :n], axis=0)
        print(f"  wave {wv} (wn={wv & 3}, wh={wv >> 2}): {n} segments, median cycles per segment:", " ".join(f"{v:5.0f}" for v in d), f" total {d.sum():.0f}")
