"""Small batches (the reference's own batch size is 32: train_regression-grf_msgn.py:93): a step is a handful of launches whose cost is launch latency.
Times, per batch size: the flat one-call step eager / replayed from one HIP graph, and the wrapper's training_step + backward + FlatAdam.step eager /
captured in one graph (static batch tensors)."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth, wrappers
from morphsym_hgnn_amd.checkpoint import load_into
dev = torch.device("cuda", 0)
spec = bench.build_spec(3)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")


def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


for B in [int(b) for b in (sys.argv[1:] or ["32", "256", "2048"])]:
    res = {"B": B}
    e = eng.Engine(spec, "bf16", device=dev)
    x, y = bench.make_batch(spec, B, 5)
    xs = e.cast_inputs(x); yd = y.to(dev)
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    gflat = torch.empty_like(flat); out = torch.empty(B * 4, 3, dtype=torch.float32, device=dev); loss = torch.empty(1, device=dev)
    step = lambda: e.step_mse(xs, flat, yd, B, out=out, grad_flat=gflat, loss=loss)
    res["flat_eager_ms"] = timeit(step)
    g = capture(step); res["flat_graph_ms"] = timeit(g.replay)
    # the wrapper surface, fp64 device inputs (the reference's convention)
    prev = torch.get_default_dtype(); torch.set_default_dtype(torch.float64)
    try:
        x64 = {k: v.to(dev, torch.float64) for k, v in x.items()}
        ei = spec.topology.edge_index_dict(B, device=dev)
        os.environ["MSHGNN_DTYPE"] = "bf16"
        w = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, spec.num_layers, spec.topology.metadata(), types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei),
                                           lr=1e-4, symmetry_mode="MorphSym", group_operator_path=cfg)
        load_into(w.model, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
        w.model.set_precision("bf16"); w.to(dev)
        opt = w.configure_optimizers()
        batch = types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei, y=y.to(dev, torch.float64).view(B, -1), batch_size=B)

        def wstep():
            opt.zero_grad(set_to_none=True)
            l = w.training_step(batch, 0)
            l.backward()
            opt.step()
        res["wrapper_eager_ms"] = timeit(wstep, 100)
        try:
            gw = capture(wstep); res["wrapper_graph_ms"] = timeit(gw.replay, 100)
        except Exception as ex:
            res["wrapper_graph_error"] = repr(ex)[:300]
    finally:
        torch.set_default_dtype(prev)
    print(res, flush=True)
