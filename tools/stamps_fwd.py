"""In-kernel phase timing of k_stack_fwd (clock64 stamps by thread 0 of every workgroup): where a tile's time goes."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3); B = 8192
stamps = torch.zeros(512 * 32, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS"] = hex(stamps.data_ptr())
e = eng.Engine(spec, sys.argv[1] if len(sys.argv) > 1 else "bf16", device=dev)      # needs a build with EXTRA=-DMSHGNN_FS_STAMPS (MSHGNN_LIB=...)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
for _ in range(3): e.forward(xs, flat, B, training=True)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(512, 32).astype(np.float64)
t0 = s[:, 0].min()
names = {0: "start", 1: "tile staged", 30: "end"}
for l in range(3):
    names.update({2 + 4 * l: f"L{l} MAC start", 3 + 4 * l: f"L{l} MAC end", 4 + 4 * l: f"L{l} barrier", 5 + 4 * l: f"L{l} epilogue end"})
order = [0, 1]
for l in range(3): order += [2 + 4 * l, 3 + 4 * l, 4 + 4 * l, 16 + l, 5 + 4 * l]
order.append(30)
for l in range(3): names[16 + l] = f"L{l} base MLP end"
first = np.argsort(s[:, 0])[:256]; second = np.argsort(s[:, 0])[256:]
for grp, idx in (("first round", first), ("second round", second)):
    print(grp, "start offsets (cycles) min/median/max:", *(int(v) for v in np.percentile(s[idx, 0] - t0, [0, 50, 100])))
    prev = 0
    for k in order:
        d = np.median(s[idx, k] - s[idx, prev]) if k else 0
        print(f"  {names[k]:18s} +{d:9.0f} cycles (median)   since start {np.median(s[idx, k] - s[idx, 0]):9.0f}")
        prev = k
print("kernel span (cycles):", int(s[:, 30].max() - t0), " clock64 ticks; at ~100 MHz wall? check: span/108us")
