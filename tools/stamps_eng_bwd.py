"""In-kernel phase timing of the backward stack kernel that the environment selects (MSHGNN_SLAB2=2 / MSHGNN_WIDE=2: k_eng_bwd; default: k_slab_bwd), build
with EXTRA=-DMSHGNN_FS_STAMPS: clock64 stamps by thread 0 of every workgroup."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3); B = 8192
stamps = torch.zeros(512 * 32, dtype=torch.int64, device=dev)
os.environ["MSHGNN_STAMPS_BWD"] = hex(stamps.data_ptr())
e = eng.Engine(spec, "bf16", device=dev)
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
y = torch.randn(B * 12, generator=g).to(dev)
for _ in range(3): e.step_mse(xs, flat, y, B)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(512, 32).astype(np.float64); s = s[s[:, 16] > 0]
names = {0: "start", 1: "tile staged"}; order = [0, 1]
for i in range(3):
    l = 2 - i
    names.update({2 + 5 * i: f"L{l} layer top", 3 + 5 * i: f"L{l} base MLP bwd", 4 + 5 * i: f"L{l} MAC engine", 5 + 5 * i: f"L{l} barrier", 6 + 5 * i: f"L{l} epilogue"})
    order += [2 + 5 * i, 3 + 5 * i, 4 + 5 * i, 5 + 5 * i, 6 + 5 * i]
prev = 0
for k in order:
    d = np.median(s[:, k] - s[:, prev]) if k else 0
    print(f"  {names[k]:18s} +{d:9.0f} cycles (median)   since start {np.median(s[:, k] - s[:, 0]):9.0f}")
    prev = k
print("workgroups:", len(s), " span:", int(s[:, 16].max() - s[:, 0].min()))
