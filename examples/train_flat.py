#!/usr/bin/env python3
"""End-to-end training loop on the flat C-ABI interface, everything on the MI355X:

    window indices --(mshgnn_step_mse_series: the window gather fused into the encoder, forward, fused MSE backward)--> flat gradient
    --(RCCL all-reduce when launched with torchrun)--> Adam on the flat buffers, step metrics on device.
(`--two-call`: mshgnn_assemble_windows, then mshgnn_step_mse -- the same bits, one more pass over the batch's windows; the only route of the
fp32 plan.)

It is what `train_model` (gnnLightning.py:1230-1400) does through Lightning + PyG's DataLoader, reduced to the hot path;
data are synthetic (a smooth random sequence whose GRFs are a fixed linear function of the joint torques, so the loss has
something to learn).  Usage:  python examples/train_flat.py [--steps 200] [--batch 8192] [--dtype bf16]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (build_spec)
from morphsym_hgnn_amd import engine as eng, synth  # noqa: E402
from morphsym_hgnn_amd.metrics import StepMetrics  # noqa: E402
from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe  # noqa: E402


def synthetic_sequence(n_rows: int, seed: int = 0):
    """A smooth random sequence; `seed` varies the DATA only -- the torque -> GRF map every rank regresses is drawn from a
    fixed generator, so that all ranks of a data-parallel run fit the same target function."""
    rng = np.random.default_rng(seed)
    label_rng = np.random.default_rng(20240131)
    smooth = lambda c: np.cumsum(rng.normal(size=(n_rows, c)), axis=0) / np.sqrt(np.arange(1, n_rows + 1))[:, None]
    seq = {"imu_acc": smooth(3), "imu_omega": smooth(3), "q": smooth(12), "qd": smooth(12), "tau": smooth(12),
           "r_o": np.tile([0.0, 0.0, 0.0, 1.0], (n_rows, 1))}
    seq["F"] = seq["tau"] @ label_rng.normal(size=(12, 12)) * 0.5 + 0.1 * seq["q"]
    return seq


def train(steps=200, batch=8192, dtype="bf16", layers=3, lr=1e-3, rows=100_000, log_every=50, quiet=False, fused_gather=None):
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    spec = bench.build_spec(layers)
    e = eng.Engine(spec, dtype=dtype, device=dev)
    store = SequenceStore(synthetic_sequence(rows, seed=rank), quadsdk_a1_c2_recipe(range(12), range(4), 150, 3), dtype=dtype, device=dev)
    flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    gflat, m, v = torch.empty_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    out = torch.empty(batch * 4, 3, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    metrics = StepMetrics(regression=True, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    losses = []
    fused_gather = (dtype == "bf16") if fused_gather is None else fused_gather
    warm = 5       # untimed: code-object load, workspace allocation, RCCL channel set-up
    for step in range(1, steps + warm + 1):
        if step == warm + 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        starts = torch.randint(0, len(store), (batch,), generator=gen, device=dev)
        if fused_gather:     # the encoder gathers its inputs from the resident series; forward + MSE + backward in the same call
            xs, y, *_ = e.step_mse_series(store, starts, flat, out=out, grad_flat=gflat, loss=loss)
        else:
            xs, y, _ = store.assemble(starts, reuse_buffers=True)
            e.step_mse(xs, flat, y.view(-1), batch, out=out, grad_flat=gflat, loss=loss)     # forward + MSE + backward in one call
        if dist is not None:
            dist.all_reduce(gflat, op=dist.ReduceOp.AVG)      # DDP semantics: mean over ranks
        e.adam_step(flat, gflat, m, v, step, lr)
        if step % log_every == 0 or step == 1 or step == steps + warm:
            metrics.calculate_losses_step(y, out.view(batch, 12))
            losses.append((step, float(loss.item()), float(metrics.rmse_loss.item())))
            if rank == 0 and not quiet:
                print(f"step {step:5d}  mse {losses[-1][1]:.5f}  rmse {losses[-1][2]:.5f}")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rank == 0 and not quiet:
        print(f"{steps} steps of {batch} windows x {world} GPU(s): {world * batch * steps / dt / 1e6:.2f} M windows/s end to end "
              f"(assembly + fwd + MSE + bwd + Adam), {dt / steps * 1e3:.3f} ms/step")
    if dist is not None:
        dist.destroy_process_group()
    return losses, world * batch * steps / dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200); ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"]); ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3); ap.add_argument("--two-call", action="store_true")
    a = ap.parse_args()
    train(a.steps, a.batch, a.dtype, a.layers, a.lr, fused_gather=False if a.two_call else None)
