"""Forward stack kernel time with and without the training stashes (what the stash stores cost), slab / slab2 / wide / 8-wave kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(int(sys.argv[1]) if len(sys.argv) > 1 else 3); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
g = torch.Generator().manual_seed(0)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
for mode, (wide, slab2, slab) in {"wide": ("2", "0", "2"), "slab2": ("0", "2", "2"), "slab": ("0", "0", "2"), "8wave": ("0", "0", "0")}.items():
    os.environ["MSHGNN_WIDE"] = wide; os.environ["MSHGNN_SLAB2"] = slab2; os.environ["MSHGNN_SLAB"] = slab
    e = eng.Engine(spec, "bf16", device=dev)
    xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    for training in (True, False):
        for _ in range(5): e.forward(xs, flat, B, training=training)
        e.profile(True)
        for _ in range(30): e.forward(xs, flat, B, training=training)
        torch.cuda.synchronize()
        st = {r["name"]: 1e3 * r["total_ms"] / r["launches"] for r in e.profile_read() if r["launches"]}
        e.profile(False)
        print(f"{mode:6s} training={training!s:5s}", {k: round(v, 1) for k, v in st.items()})
