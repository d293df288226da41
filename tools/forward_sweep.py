"""Evaluation forward (mshgnn_forward, training=0) over batch sizes, 8-wave stack kernel against the slab one, us per launch sequence.
usage: python tools/forward_sweep.py "32 2048 4096 4112 6144 8192" [layers] [dtype] [config]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 5 and sys.argv[5] == "child":
    import torch
    import bench
    from morphsym_hgnn_amd import engine as eng, synth
    B, L, dtype, config = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dev = torch.device("cuda:0")
    spec = bench.build_spec(L, config, 128)
    e = eng.Engine(spec, dtype, device=dev)
    x, y = bench.make_batch(spec, B, 1)
    xs = e.cast_inputs({k: v.to(dev) for k, v in x.items()})
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    out = torch.empty(B * spec.num_nodes[spec.out_type], spec.out_channels, dtype=torch.float32, device=dev)
    for tr in (False, True):
        for _ in range(20):
            e.forward(xs, flat, B, training=tr, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            e.forward(xs, flat, B, training=tr, out=out)
        torch.cuda.synchronize()
        print(f"{(time.perf_counter() - t0) / 200 * 1e6:.1f}", end=" ")
    gout = torch.full_like(out, 1.0 / out.numel()); g = torch.empty_like(flat)
    for _ in range(20):
        e.forward(xs, flat, B, training=True, out=out); e.backward(xs, flat, gout, B, grad_flat=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        e.forward(xs, flat, B, training=True, out=out); e.backward(xs, flat, gout, B, grad_flat=g)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 200 * 1e6:.1f}", end=" ")
    print()
    sys.exit(0)

L = sys.argv[2] if len(sys.argv) > 2 else "3"
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
config = sys.argv[4] if len(sys.argv) > 4 else "a1c2"
print("B: us (eval forward, training forward, training forward + backward) per route: default (compile-time program where the plan has one) | MSHGNN_SLAB=0 (8-wave kernels) | MSHGNN_SPEC=0 (interpreting kernels, round 5's routing) | MSHGNN_SPEC=0 MSHGNN_SLAB=2", flush=True)
for b in sys.argv[1].split():
    row = {}
    for name, over in (("default", {}), ("8wave", {"MSHGNN_SLAB": "0"}), ("interp", {"MSHGNN_SPEC": "0"}), ("interp_slab", {"MSHGNN_SPEC": "0", "MSHGNN_SLAB": "2"})):
        env = dict(os.environ)
        env.pop("MSHGNN_SLAB", None); env.pop("MSHGNN_SPEC", None)
        env.update(over)
        p = subprocess.run([sys.executable, __file__, b, L, dtype, config, "child"], capture_output=True, text=True, env=env)
        row[name] = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-200:]
    print("B", b, "tiles", (int(b) + 15) // 16, " | ".join(f"{k} {v}" for k, v in row.items()), flush=True)
