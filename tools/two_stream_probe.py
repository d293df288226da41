"""Go / no-go probe: do two half-batch steps on two HIP streams beat one full-batch step?

The step's launches run strictly in series: enc (HBM-bound) -> stack_step (latency-bound, 2.3 TB/s, 18 % MFMA) -> gradw (HBM-bound) -> finalize.
If the stack launch of one half-batch can run beside the HBM-bound launches of the other, the pair should finish sooner than the serial sum.
Existing kernels, existing entry points; two Engines (two workspaces), two streams.

  python tools/two_stream_probe.py [--trace]      (MSHGNN_SLAB is set per engine through the environment at plan creation)

Modes timed (ms per 8192 windows):
  full            one mshgnn_step_mse of 8192 windows
  halves_serial   two steps of 4096 on ONE stream
  halves_joined   two steps of 4096 on two streams, streams joined after every pair (what a training step needs: one optimizer update per pair)
  halves_free     the two streams free-running (steady-state pipeline, upper bound of what an offset schedule could reach)
  halves_offset   stream B's step starts when stream A's FORWARD launch sequence is done (forward / backward_mse two-call route, event between them)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from morphsym_hgnn_amd import engine as eng, synth  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.build_spec(3)


def mk(B, seed, slab):
    if slab is None:
        os.environ.pop("MSHGNN_SLAB", None)
    else:
        os.environ["MSHGNN_SLAB"] = str(slab)
    e = eng.Engine(spec, dtype="bf16", device=dev)
    x, y = bench.make_batch(spec, B, seed)
    xs = e.cast_inputs(x)
    y = y.to(dev)
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    gflat = torch.empty_like(flat)
    out = torch.empty(B * 4, 3, dtype=torch.float32, device=dev)
    loss = torch.empty(1, device=dev)

    def step():
        e.step_mse(xs, flat, y, B, out=out, grad_flat=gflat, loss=loss)

    def fwd():
        e.forward(xs, flat, B, training=True, out=out)

    def bwd():
        e.backward_mse(xs, flat, out, y, B, grad_flat=gflat, loss=loss)

    step.fwd, step.bwd, step.engine = fwd, bwd, e
    return step


def timeit(fn, n=60, reps=5):
    for _ in range(8):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    trace = "--trace" in sys.argv
    n = 6 if trace else 60
    reps = 1 if trace else 5
    res = {"what": __doc__.split("\n")[0], "workload": "A1-C2 h=128 L=3 bf16 plan, 8192 windows per pair", "ms_per_8192_windows": {}}
    r = res["ms_per_8192_windows"]
    full = mk(8192, 1, None)
    r["full"] = timeit(full, n, reps)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for slab, tag in ((2, "slab_1wg_per_cu"), (0, "eight_wave")):
        a, b = mk(4096, 2, slab), mk(4096, 3, slab)
        r[f"{tag}.one_half"] = timeit(a, n, reps)

        def serial():
            a(); b()
        r[f"{tag}.halves_serial"] = timeit(serial, n, reps)

        def joined():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                a()
            with torch.cuda.stream(s2):
                b()
            cur.wait_stream(s1); cur.wait_stream(s2)
        r[f"{tag}.halves_joined"] = timeit(joined, n, reps)

        def free():
            with torch.cuda.stream(s1):
                a()
            with torch.cuda.stream(s2):
                b()
        r[f"{tag}.halves_free"] = timeit(free, n, reps)

        def offset():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                a.fwd()
                ev = torch.cuda.Event(); ev.record(s1)
                a.bwd()
            with torch.cuda.stream(s2):
                s2.wait_event(ev)
                b.fwd(); b.bwd()
            cur.wait_stream(s1); cur.wait_stream(s2)
        r[f"{tag}.halves_offset_two_call"] = timeit(offset, n, reps)

        def two_call_serial():
            a.fwd(); a.bwd(); b.fwd(); b.bwd()
        r[f"{tag}.halves_serial_two_call"] = timeit(two_call_serial, n, reps)
        del a, b
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
