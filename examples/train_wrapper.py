#!/usr/bin/env python3
"""The reference's training loop through its own wrapper API, on the MI355X:

    SequenceStore.batch (window indices; the encoder gathers them from the resident series -- `--assemble`: a separate gather pass first)
    ->  HGNN_C2_Lightning_Reg.training_step(batch)  ->  loss.backward()
    ->  the optimizer `configure_optimizers()` returns (optim.FlatAdam: a torch.optim.Adam whose step is one launch on the flat buffers)

i.e. what Lightning's Trainer does with the reference's `HGNN_C2_Lightning_Reg` (gnnLightning.py:564-778, train_model :1230-1400), driven
by hand because `lightning` is not installed here -- same method calls in the same order (training_step, zero_grad, backward, step).
Data are synthetic (examples/train_flat.py's sequence: GRFs are a fixed linear function of the joint torques).
Usage:  python examples/train_wrapper.py [--steps 200] [--batch 8192] [--dtype bf16]   (under torchrun: one rank per GPU, `ddp.flat_data_parallel`)
"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (build_spec)
from examples.train_flat import synthetic_sequence  # noqa: E402
from morphsym_hgnn_amd import wrappers  # noqa: E402
from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe  # noqa: E402


def train(steps=200, batch=8192, dtype="bf16", layers=3, lr=1e-3, rows=100_000, log_every=50, quiet=False, assemble=False):
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:      # one process per GPU (torchrun): every rank trains on its own windows, one all-reduce of the flat gradient per step
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    spec = bench.build_spec(layers)
    store = SequenceStore(synthetic_sequence(rows, seed=rank), quadsdk_a1_c2_recipe(range(12), range(4), 150, 3), dtype=dtype, device=dev)
    ei = spec.topology.edge_index_dict(batch, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def next_batch():      # a WindowBatch: window indices only -- training_step lets the encoder gather them; `assemble=True`: tensors first
        starts = torch.randint(0, len(store), (batch,), generator=gen, device=dev)
        wb = store.batch(starts, ei)
        if assemble:
            wb.x_dict
        return wb

    os.environ["MSHGNN_DTYPE"] = dtype
    torch.manual_seed(0)
    cfg = os.path.join(bench.ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")
    # the lazy-initialising dummy forward (gnnLightning.py:593-595) must see the reference's feature widths (900 / 450 / 1), not the
    # aligned pitch of assembled batches: the encoder's in-features are read off it
    first = next_batch()
    dummy = types.SimpleNamespace(edge_index_dict=ei, x_dict={t: x[:, :store.recipe.width(t)].float().contiguous() for t, x in first.x_dict.items()})
    model = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, layers, spec.topology.metadata(), dummy, optimizer="adam", lr=lr,
                                           symmetry_mode="MorphSym", group_operator_path=cfg, grf_body_to_world_frame=False).to(dev)
    if dist is not None:
        from morphsym_hgnn_amd import ddp
        with torch.no_grad():
            model.model(x_dict=dict(first.x_dict), edge_index_dict=ei)      # parameters become views of the flat buffer
        ddp.flat_data_parallel(model)                                      # broadcast rank 0's parameters; training_step all-reduces the flat gradient
    opt = model.configure_optimizers()
    losses = []
    warm = 5
    for step in range(1, steps + warm + 1):
        if step == warm + 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = model.training_step(next_batch(), step)       # Lightning's order: training_step, zero_grad, backward, optimizer step
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if step % log_every == 0 or step == 1 or step == steps + warm:
            losses.append((step, float(model.logged["train_MSE_loss"]), float(model.logged["train_RMSE_loss"]), float(model.logged["train_L1_loss"])))
            if not quiet and rank == 0:
                print(f"step {step:5d}  mse {losses[-1][1]:.5f}  rmse {losses[-1][2]:.5f}  l1 {losses[-1][3]:.5f}")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if dist is not None:
        dist.destroy_process_group()
    if not quiet and rank == 0:
        print(f"{steps} steps of {batch} windows x {world} GPU(s): {world * batch * steps / dt / 1e6:.2f} M windows/s end to end (window assembly + training_step + backward + "
              f"{type(opt).__name__}.step), {dt / steps * 1e3:.3f} ms/step")
    return losses, world * batch * steps / dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200); ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "x3"]); ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3); ap.add_argument("--assemble", action="store_true")
    a = ap.parse_args()
    train(a.steps, a.batch, a.dtype, a.layers, a.lr, assemble=a.assemble)
