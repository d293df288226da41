"""Does the chip have room for two half-batch steps side by side?  Two engines (own workspaces) on two streams, each stepping B/2 windows, against one engine stepping B:
aggregate windows/s.  The half steps' issue-bound stack launch and HBM-bound weight-gradient launch can then share CUs.  usage: python tools/two_stream_probe.py [B] [layers] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from morphsym_hgnn_amd import engine as eng, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
spec = bench.build_spec(L, "a1c2", 128)


def setup(b, seed):
    e = eng.Engine(spec, dtype, device=dev)
    x, y = bench.make_batch(spec, b, seed)
    xs = e.cast_inputs({k: v.to(dev) for k, v in x.items()})
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    yd = y.reshape(-1).to(dev, torch.float32)
    out = torch.empty(b * spec.num_nodes[spec.out_type], spec.out_channels, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev); g = torch.empty_like(flat)
    return lambda: e.step_mse(xs, flat, yd, b, out=out, loss=loss, grad_flat=g)


def timed(fns, streams, n=200):
    for _ in range(20):
        for f, s in zip(fns, streams):
            with torch.cuda.stream(s):
                f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f, s in zip(fns, streams):
            with torch.cuda.stream(s):
                f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


one = timed([setup(B, 1)], [torch.cuda.Stream(dev)])
print(f"one stream,  B={B}: {one * 1e3:.4f} ms/step = {B / one / 1e6:.2f} M windows/s", flush=True)
for parts in (2, 4):
    fns = [setup(B // parts, 10 + i) for i in range(parts)]
    t = timed(fns, [torch.cuda.Stream(dev) for _ in range(parts)])
    print(f"{parts} streams, B={B // parts} each: {t * 1e3:.4f} ms per round = {B / t / 1e6:.2f} M windows/s", flush=True)
    t1 = timed(fns, [torch.cuda.current_stream(dev)] * parts)
    print(f"   (the same {parts} steps on one stream: {t1 * 1e3:.4f} ms per round = {B / t1 / 1e6:.2f} M windows/s)", flush=True)
