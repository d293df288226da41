"""In-kernel phase timing of k_slab_step (build with EXTRA=-DMSHGNN_FS_STAMPS, run with MSHGNN_LIB=<that build>): clock64 stamps by thread 0 of
every workgroup, forward sweep in slots [0, 32) and backward sweep in the second half of the buffer.  Two workgroups share a CU, so a phase's
duration includes what the neighbour takes from the same SIMDs / L1 / LDS.  Usage: python tools/stamps_slab_step.py [layers] [windows]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda", 0)
spec = bench.build_spec(L)
G = (B + 15) // 16
stamps = torch.zeros(G * 64, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS"] = hex(stamps.data_ptr())
e = eng.Engine(spec, "bf16", device=dev)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
y = torch.randn(B * 12, generator=g).to(dev)
for _ in range(3): e.step_mse(xs, flat, y, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
f = s[:G * 32].reshape(G, 32); b = s[G * 32:].reshape(G, 32)
rows = [("start", f[:, 0]), ("tile staged", f[:, 1])]
for l in range(L):
    rows += [(f"F L{l} group A", f[:, 2 + 4 * l]), (f"F L{l} group B (+ base MLP)", f[:, 3 + 4 * l]), (f"F L{l} barrier", f[:, 4 + 4 * l]), (f"F L{l} stores + barrier", f[:, 5 + 4 * l])]
rows += [("tail: operands / entry", f[:, 24]), ("tail: decoder + loss + dX_L passes", f[:, 25]), ("tail: row shuffles", f[:, 26]), ("tail: partials to LDS + barriers", f[:, 27]),
         ("tail: slab sums written (end)", f[:, 30]), ("B start (residual rows read)", b[:, 0])]
for i in range(L):
    l = L - 1 - i
    rows += [(f"B L{l} mask + barrier", b[:, 1 + 6 * i]), (f"B L{l} base MLP chain", b[:, 2 + 6 * i]), (f"B L{l} group A", b[:, 3 + 6 * i]), (f"B L{l} group B", b[:, 4 + 6 * i]),
             (f"B L{l} barrier", b[:, 5 + 6 * i]), (f"B L{l} stores + barrier", b[:, 6 + 6 * i])]
prev = rows[0][1]
tot = np.median(rows[-1][1] - rows[0][1])
for nm, v in rows:
    d = np.median(v - prev)
    print(f"  {nm:32s} +{d:9.0f} cycles ({100 * d / tot:5.1f} %)   since start {np.median(v - rows[0][1]):9.0f}")
    prev = v
dur = rows[-1][1] - rows[0][1]
print(f"tile duration: min {dur.min():.0f}  median {np.median(dur):.0f}  p90 {np.percentile(dur, 90):.0f}  max {dur.max():.0f}   (clock64 ticks at 100 MHz x ... see s_memtime; compare ratios)")
print(f"start spread {f[:, 0].max() - f[:, 0].min():.0f}")
