"""In-kernel phase timing of k_slab_fwd (build with EXTRA=-DMSHGNN_FS_STAMPS): clock64 stamps by thread 0 of every workgroup.
Two workgroups share a CU, so a phase's duration includes what the neighbour takes from the same SIMDs / L1 / LDS."""
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
for _ in range(3): e.step_mse(xs, flat, y, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(512, 32).astype(np.float64)
names = {0: "start", 1: "tile staged", 30: "end (decoder + MSE tail)"}; order = [0, 1]
for l in range(3):
    names.update({2 + 4 * l: f"L{l} group A (joints)", 3 + 4 * l: f"L{l} group B (+ base MLP)", 4 + 4 * l: f"L{l} barrier", 5 + 4 * l: f"L{l} stores + barrier"})
    order += [2 + 4 * l, 3 + 4 * l, 4 + 4 * l, 5 + 4 * l]
order.append(30)
prev = 0
for k in order:
    d = np.median(s[:, k] - s[:, prev]) if k else 0
    print(f"  {names[k]:28s} +{d:9.0f} cycles (median)   since start {np.median(s[:, k] - s[:, 0]):9.0f}")
    prev = k
print("start spread (cycles):", int(s[:, 0].max() - s[:, 0].min()), " span:", int(s[:, 30].max() - s[:, 0].min()))
# distribution over the workgroups: the launch lasts as long as its slowest tile, not the median one
dur = s[:, 30] - s[:, 0]; st = s[:, 0] - s[:, 0].min(); en = s[:, 30] - s[:, 0].min()
q = lambda a, p: float(np.percentile(a, p))
print(f"tile duration: min {dur.min():.0f}  median {np.median(dur):.0f}  p90 {q(dur, 90):.0f}  p99 {q(dur, 99):.0f}  max {dur.max():.0f}")
print(f"tile start   : median {np.median(st):.0f}  p90 {q(st, 90):.0f}  max {st.max():.0f}")
print(f"tile end     : median {np.median(en):.0f}  p90 {q(en, 90):.0f}  max {en.max():.0f}   (clock64 is per XCD: differences across XCDs include their clock offsets)")
xcd = np.arange(512) % 8
for x in range(8):
    m = xcd == x
    print(f"  XCD {x}: start {np.median(s[m, 0] - s[m, 0].min()):7.0f}  duration median {np.median(dur[m]):7.0f} max {dur[m].max():7.0f}  last end - first start {s[m, 30].max() - s[m, 0].min():7.0f}")
for k, nm in ((1, "tile staged"), (5, "L0 end"), (9, "L1 end"), (13, "L2 end")):
    d = s[:, k] - s[:, 0]
    print(f"  since start at '{nm}': median {np.median(d):.0f}  p99 {q(d, 99):.0f}  max {d.max():.0f}")
