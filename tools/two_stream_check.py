"""Experiment: does running two half-batch chains on two HIP streams beat one full-batch chain?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
dev = torch.device("cuda", 0)
spec = bench.build_spec(3)
def mk(B, seed):
    e = eng.Engine(spec, dtype="bf16", device=dev)
    g = torch.Generator().manual_seed(seed)
    imu = torch.randn(B, 1, 900, generator=g)
    x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
    xs = e.cast_inputs(x); y = torch.randn(B * 12, generator=g).to(dev)
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    gflat = torch.empty_like(flat); out = torch.empty(B * 4, 3, dtype=torch.float32, device=dev); loss = torch.empty(1, device=dev)
    def step():
        e.forward(xs, flat, B, training=True, out=out)
        e.backward_mse(xs, flat, out, y, B, grad_flat=gflat, loss=loss)
    return step
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = mk(8192, 1)
print("one chain B=8192: %.4f ms" % timeit(full))
for nsplit in (2, 4):
    Bh = 8192 // nsplit
    steps = [mk(Bh, 2 + i) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    print("single chain B=%d: %.4f ms" % (Bh, timeit(steps[0])))
    def both():
        for s, st in zip(steps, streams):
            with torch.cuda.stream(st): s()
    print("%d chains B=%d on %d streams: %.4f ms" % (nsplit, Bh, nsplit, timeit(both)))
