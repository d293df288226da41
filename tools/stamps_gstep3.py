"""Phase clocks of k_gstep3 (build with EXTRA=-DG3_STAMPS): per workgroup [setup, chunk loop, of which inside the asm MFMA block, epilogue, chunks, total]; the
buffer holds the LAST launch of the step that used the kernel (a backward layer)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphsym_hgnn_amd import engine as eng, synth, topology
from morphsym_hgnn_amd.spec import ModelSpec
B = 1024
spec = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
os.environ["MSHGNN_G3_STAMPS"] = hex(stamps.data_ptr()); os.environ["MSHGNN_GEN_TILE"] = sys.argv[1] if len(sys.argv) > 1 else "4"
n_y = spec.out_channels * spec.num_nodes[spec.out_type]
x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, n_y)
e = eng.Engine(spec, "bf16")
xs = e.cast_inputs(x_dict); flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
for _ in range(2): e.forward(xs, flat, B, training=False)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64); s = s[s[:, 5] > 0]
for k, nm in enumerate(["setup", "chunk loop", "  inside asm", "epilogue", "chunks", "total", "  stage", "  to barrier"]):
    print(f"{nm:14s} median {np.median(s[:, k]):9.0f}  p90 {np.percentile(s[:, k], 90):9.0f}  max {s[:, k].max():9.0f}")
print("workgroups", len(s), " asm cycles per chunk", np.median(s[:, 2] / s[:, 4]))
