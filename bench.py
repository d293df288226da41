#!/usr/bin/env python3
"""bench.py -- graph-windows/sec, fwd + MSE loss + bwd, A1-C2 GRF regression (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one pass of the hot path over one minibatch of synthetic A1-shaped windows that is already
resident in HBM: forward (encoder, L message-passing layers, decoder) + MSE loss + backward with every
parameter gradient materialised (mshgnn_step_mse; + the RCCL gradient all-reduce when N > 1).  Prints ONE JSON line on rank 0.

The workload is BASELINE.json configs[1]: A1 C2 GRF regression, h=128, L=3, grf_dimension=3, batch 8192
time-windows per GPU (weak scaling).  `roofline` is computed for the kernel with the largest share of the
step, from HIP events recorded around every kernel by the C-ABI (mshgnn_profile_*), in a second pass over
the same K steps (so the events do not perturb `value`).  `cpu_baseline` times the fp64 oracle (a port of
the reference's CPU path, oracle/ms_hgnn_oracle.py) on the host cores on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {  # /opt/skills/guides/MI355X_MICROARCH.md: chip-level parameters
    "hbm_GBs": 8000.0,
    "mfma_TFLOPs": {"f32": 157.3, "bf16": 2500.0},
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8192, help="windows per GPU")
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--dtype", default=os.environ.get("MSHGNN_BENCH_DTYPE", "bf16"), choices=["f32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=1024)
    return ap.parse_args()


def build_spec(layers):
    import yaml
    from morphsym_hgnn_amd import synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    with open(os.path.join(ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")) as f:
        group = yaml.safe_load(f)
    return ModelSpec(kind="c2", topology=topology.a1_c2(), hidden=128, num_layers=layers,
                     widths=synth.feature_widths("c2", True), regression=True, grf_dimension=3, group=group)


def cpu_baseline(spec, batch, budget_s=24.0):
    """Oracle (port of the reference CPU path, fp64) timed on the host cores: fwd + MSE + bwd.  The thread count
    is picked from a short scan (small-matrix torch code does not scale to every core of the host)."""
    from morphsym_hgnn_amd import synth
    from oracle import ms_hgnn_oracle as orc
    cfg = orc.OracleConfig(kind="c2", num_layers=spec.num_layers, edge_types=spec.edge_types, regression=True,
                           grf_dimension=3, group=spec.group)
    x_dict, y = synth.make_windows(1, batch, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(1, spec.param_shapes())
    ei = spec.topology.edge_index_dict(batch)

    def one():
        t0 = time.perf_counter()
        orc.step(cfg, params, x_dict, ei, y, batch)
        return time.perf_counter() - t0

    default_threads = torch.get_num_threads()
    best = (float("inf"), default_threads)
    t_start = time.perf_counter()
    for n in sorted({8, 16, 32, 64, default_threads}):
        if n > default_threads or time.perf_counter() - t_start > budget_s / 2:
            continue
        torch.set_num_threads(n)
        one()
        t = min(one(), one())
        if t < best[0]:
            best = (t, n)
    torch.set_num_threads(best[1])
    times = []
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 40):
        times.append(one())
    torch.set_num_threads(default_threads)
    times.sort()
    med = times[len(times) // 2]
    return {"value": batch / med, "unit": "windows/s", "cores": best[1], "kind": "port",
            "sample": f"oracle fp64 fwd+MSE+bwd, A1-C2 h128 L{spec.num_layers}, B={batch}, median of {len(times)} steps, "
                      f"{best[1]} threads (best of a scan; host has {os.cpu_count()} cpus)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched through torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1 or os.environ.get("MSHGNN_BENCH_FORCE_DIST") == "1":   # (FORCE_DIST: a 1-rank RCCL group, to exercise the N > 1 code path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from morphsym_hgnn_amd import engine as eng, synth
    spec = build_spec(args.layers)
    e = eng.Engine(spec, dtype=args.dtype, device=device)
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    # synthetic A1-shaped windows: one IMU window tiled to both base nodes, joints ~N(0,1), foot ones
    imu = torch.randn(B, 1, 900, generator=g)
    x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g),
         "foot": torch.ones(B * 4, 1)}
    xs = e.cast_inputs(x)
    y = torch.randn(B * 12, generator=g).to(device)
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), device)
    gflat = torch.empty_like(flat)
    out = torch.empty(B * 4, 3, dtype=torch.float32, device=device)

    loss_buf = torch.empty(1, dtype=torch.float32, device=device)

    split = int(e.info.grad_split)
    # two-phase step + interleaved all-reduce: OFF by default.  Measured with a 1-rank RCCL group (no wire time): the split
    # launches and the extra stream hand-offs cost +51 us per step, about what an 8-GPU all-reduce of 3.3 MB could hide, and
    # more than a 2- or 4-GPU one could -- the plain sequence is at least as fast at this gradient size (4 MB).
    overlap = dist is not None and split > 0 and os.environ.get("MSHGNN_BENCH_OVERLAP", "0") == "1"

    def step():
        # forward + MSE + backward through the C-ABI (on the bf16 plan the decoder, the loss and the decoder backward run inside
        # the fused forward kernel; == mshgnn_forward + mshgnn_backward_mse, checked by tests/test_engine_gpu.py)
        if not overlap:
            _, loss, _ = e.step_mse(xs, flat, y, B, out=out, grad_flat=gflat, loss=loss_buf)
            if dist is not None:
                dist.all_reduce(gflat)   # RCCL sum over ranks (DDP semantics: mean = sum / world, folded into lr)
            return loss
        # N > 1: the same step in two calls; the all-reduce of everything but the encoder's gradients (83 % of the buffer) runs on
        # RCCL's stream while this stream computes the encoder's weight gradients, then the encoder's slice follows.  Both
        # collectives complete inside the step (wait() makes this stream wait for them).
        e.step_mse_phase(0, xs, flat, y, B, out, gflat, loss_buf)
        w1 = dist.all_reduce(gflat[split:], async_op=True)
        e.step_mse_phase(1, xs, flat, y, B, out, gflat, loss_buf)
        w2 = dist.all_reduce(gflat[:split], async_op=True)
        w1.wait(); w2.wait()
        return loss_buf

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * B * args.steps / dt

    # second pass with per-kernel HIP events -> roofline of the dominant kernel
    e.profile(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    stats = [s for s in e.profile_read() if s["launches"] > 0]
    e.profile(False)
    total_ms = sum(s["total_ms"] for s in stats)
    dom = max(stats, key=lambda s: s["total_ms"])
    avg_s = dom["total_ms"] / dom["launches"] * 1e-3
    # the two-phase step (N > 1) launches the weight-gradient kernel twice per step: per-launch work = per-step work / launches
    per_step = max(1, round(dom["launches"] / args.steps))
    dom = dict(dom, flops_per_window=dom["flops_per_window"] / per_step, bytes_per_window=dom["bytes_per_window"] / per_step)
    if dom["bound"] == "mfma":
        achieved = dom["flops_per_window"] * B / avg_s / 1e12
        peak = PEAK["mfma_TFLOPs"][args.dtype]
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None}
    else:
        achieved = dom["bytes_per_window"] * B / avg_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK["hbm_GBs"], "unit": "GB/s", "frac": achieved / PEAK["hbm_GBs"], "traffic": None}
    # HBM-side traffic per launch of that kernel, from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    # WRITE_SIZE runs of this same command, gfx950 corrections applied -- tools/summarize_pmc.py); null if absent
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
        if files and args.dtype == "bf16" and B == 8192 and args.layers == 3:
            pm = json.load(open(files[-1]))["kernels"]
            key = dom["name"].rstrip("0123456789")
            if key in pm:
                roof["traffic"] = pm[key]["hbm_bytes"]
                roof["traffic_source"] = os.path.basename(files[-1])
    except Exception:  # noqa: BLE001
        pass
    if roof.get("traffic") is not None:
        roof["traffic"] = roof["traffic"] / per_step
    roof["kernel"] = dom["name"]
    roof["launches_per_step"] = per_step
    roof["avg_us"] = avg_s * 1e6
    roof["share_of_step"] = dom["total_ms"] / max(total_ms, 1e-9)
    kernels = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 2) for s in stats}

    res = {
        "metric": "graph-windows/sec fwd+bwd, A1-C2 GRF regression", "value": value, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"A1-C2 GRF regression (3-D), h=128, L={args.layers}, T=150, {B} windows/GPU, "
                               f"fwd+MSE+bwd, all parameter gradients" + (", RCCL all-reduce" if world > 1 else ""),
                   "global_batch": B * world, "parallelism": f"dp{world}"},
        "roofline": roof, "kernel_us": kernels,
        "algorithmic_flops_per_window": e.info.flops_fwd + e.info.flops_bwd,
        "loss": float(loss.item()),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(spec, args.cpu_batch)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
