"""Data-parallel sharding of the MS-HGNN path across the GPUs of one node.

Windows are independent, so the path shards with no data-path collective: the global minibatch is split
contiguously across ranks (one process per GPU), weights are replicated, and the ONLY exchange per step is
one sum-all-reduce of the flat fp32 gradient buffer the C-ABI produces (3.98 MB for A1-C2 h128 L3) -- RCCL
over xGMI on GPUs (`backend="nccl"`), gloo on CPU in the tests.  This is what Lightning-DDP would do for the
reference with devices>1 (gnnLightning.py:1396-1400), with mean semantics (sum / world size).

The nn.Module surface (models.py) needs none of this: its parameters are ordinary nn.Parameters, so
torch.nn.parallel.DistributedDataParallel / Lightning DDP wrap it unchanged (their bucketed all-reduce, per-parameter autograd hooks).
`flat_data_parallel` is the one-exchange alternative for the training-step wrappers: the one-call training step keeps running under
torch.distributed and all-reduces its flat gradient buffer once before it hands it to the parameters.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) window range of `rank`; the first (global_batch % world) ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(global_batch, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_x_dict(x_dict, num_nodes, global_batch: int, rank: int, world: int):
    """Slice reference-convention inputs ([B*n_t, F_t], graph-major) to this rank's windows."""
    b, e = shard_bounds(global_batch, rank, world)
    return {t: x[b * num_nodes[t]: e * num_nodes[t]] for t, x in x_dict.items()}, (b, e)


def allreduce_gradients_(flat_grad: torch.Tensor, local_windows: int, global_windows: int, group=None) -> torch.Tensor:
    """In place: turn per-rank gradients of the LOCAL mean loss into the gradient of the GLOBAL mean loss.
    grad_global = sum_r (local_windows_r / global_windows) * grad_r  -- one all-reduce of the flat buffer."""
    import torch.distributed as dist
    flat_grad.mul_(float(local_windows) / float(global_windows))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def allreduce_mean_bf16_(flat_grad: torch.Tensor, scratch16: torch.Tensor, group=None) -> torch.Tensor:
    """Opt-in: the mean all-reduce of the flat fp32 gradient with bf16 on the wire (half the bytes per xGMI link: 2 MB instead of 4 MB at A1-C2 L=3).
    Parity cost, stated: every rank's gradient is rounded to bf16 (8 significant bits, <= 2^-9 relative per element) before the sum and the sum is
    carried in bf16 by the collective -- the exchanged gradient is within ~world x 2^-9 relative of the fp32 exchange per element, so it is NOT a
    parity-grade (1e-4) route; forward, loss and the local gradients are untouched.  scratch16: bf16, same numel, reused every step."""
    import torch.distributed as dist
    scratch16.copy_(flat_grad)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1:
        dist.all_reduce(scratch16, op=dist.ReduceOp.SUM, group=group)
    flat_grad.copy_(scratch16)
    if world > 1:
        flat_grad.div_(world)
    return flat_grad


class LiveGradientExchange:
    """Mean all-reduce of the flat gradient that moves only the elements that can be non-zero.  At a depth below the graph's diameter whole parameter
    tensors get exact-zero gradients on EVERY rank (spec.dead_parameters: A1-C2 at 3 layers 64 % of the 3.98 MB buffer), so the exchange packs the live
    elements (one gather launch), all-reduces 1.4 MB instead of 4 MB and scatters the result back (one launch); the dead elements stay the zeros the step
    wrote.  Exact -- the same values as the full exchange (`tests/test_ddp.py`).  Whether two extra launches beat 2.6 MB less on the wire depends on the
    machine: `bench.py --grad-exchange auto` times both."""

    def __init__(self, spec, device, dtype=torch.float32):
        self.index = spec.live_gradient_index().to(device)
        self.packed = torch.empty(self.index.numel(), dtype=dtype, device=device)
        self.fraction = self.index.numel() / max(1, spec.flat_size())

    def allreduce_mean_(self, flat_grad: torch.Tensor, group=None, comm=None) -> torch.Tensor:
        """comm: a `StreamAllReduce` -- the packed buffer is then exchanged on the current stream through the C-ABI instead of torch.distributed."""
        import torch.distributed as dist
        torch.index_select(flat_grad, 0, self.index, out=self.packed)
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if comm is not None:
            comm.allreduce_mean_(self.packed)
        elif world > 1:
            dist.all_reduce(self.packed, op=dist.ReduceOp.SUM, group=group)
            self.packed.div_(world)
        flat_grad.index_copy_(0, self.index, self.packed)
        return flat_grad


class StreamAllReduce:
    """The step's ONE collective enqueued on the step's own HIP stream through the C-ABI (`mshgnn_comm_*`, include/mshgnn.h): `ncclAllReduce(buf, buf, n,
    float32, avg)` goes straight onto the stream that runs the step's kernels -- no Python collective call, no hand-over to torch.distributed's side
    stream and back (measured on a 1-rank group: +16 us per step through `dist.all_reduce`), and step + exchange can be captured in one HIP graph.
    RCCL is the instance the process already carries (torch's bundled librccl.so).  The communicator is created once: rank 0 draws the 128-byte
    unique id, an already initialised torch.distributed group (any backend) carries it to the other ranks.  One process per GPU, as everywhere."""

    def __init__(self, device, group=None):
        import ctypes as C
        import os
        import torch.distributed as dist
        from . import engine as eng
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("StreamAllReduce needs an initialised torch.distributed process group to hand the communicator id around")
        self.lib = eng.load_library()
        self.device = torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._path = cand.encode() if os.path.exists(cand) else None
        # every rank first proves it can bind RCCL (drawing an id loads the library and is local); the verdicts are combined BEFORE the collective
        # ncclCommInitRank, so that one rank without a loadable librccl makes every rank raise instead of leaving the others inside the rendezvous
        raw = C.create_string_buffer(128)
        rc = self.lib.mshgnn_comm_unique_id(self._path, raw)
        err = self.lib.mshgnn_last_error().decode() if rc else ""
        oks = [None] * self.world
        dist.all_gather_object(oks, rc == 0, group=group)
        if not all(oks):
            raise RuntimeError(f"StreamAllReduce: RCCL could not be bound on rank(s) {[i for i, o in enumerate(oks) if not o]}" + (f" (this rank: {err})" if err else ""))
        ident = [raw.raw if self.rank == 0 else None]
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            eng._check(self.lib, self.lib.mshgnn_comm_create(self._path, C.create_string_buffer(ident[0], 128), self.world, self.rank, C.byref(self._comm)),
                       "mshgnn_comm_create")

    def allreduce_mean_(self, flat_grad: torch.Tensor) -> torch.Tensor:
        """In place: mean over the ranks, stream-ordered on torch's current stream of the buffer's device (where the step was launched)."""
        from . import engine as eng
        if flat_grad.dtype != torch.float32 or not flat_grad.is_contiguous() or flat_grad.device != self.device:
            raise ValueError("StreamAllReduce: a contiguous fp32 tensor on the communicator's device")
        with torch.cuda.device(self.device):
            eng._check(self.lib, self.lib.mshgnn_comm_allreduce_mean(self._comm, flat_grad.data_ptr(), flat_grad.numel(),
                                                                       torch.cuda.current_stream(self.device).cuda_stream), "mshgnn_comm_allreduce_mean")
        return flat_grad

    def allreduce_sum_(self, buf: torch.Tensor) -> torch.Tensor:
        from . import engine as eng
        with torch.cuda.device(self.device):
            eng._check(self.lib, self.lib.mshgnn_comm_allreduce_sum(self._comm, buf.data_ptr(), buf.numel(), torch.cuda.current_stream(self.device).cuda_stream),
                       "mshgnn_comm_allreduce_sum")
        return buf

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            torch.cuda.synchronize(self.device)
            self.lib.mshgnn_comm_destroy(self._comm)
            self._comm.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_STREAM_COMMS = {}      # process group (None: the default one) -> StreamAllReduce registered by flat_data_parallel(stream_collective=True)


def _sum_all_reduce_(t: torch.Tensor, group=None) -> None:
    """Sum over the ranks, in place: through the group's registered C-ABI communicator on the current stream where there is one, else torch.distributed."""
    import torch.distributed as dist
    comm = _STREAM_COMMS.get(group)
    if comm is not None and t.is_cuda and t.device == comm.device and t.dtype == torch.float32 and t.is_contiguous():
        comm.allreduce_sum_(t)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def exchange_flat_gradient_(buf: torch.Tensor, n_flat: int, local_windows: int, group=None, weight_by_windows: bool = False, live=None) -> torch.Tensor:
    """The ONE exchange of a `flat_data_parallel` step, in place on `buf` (fp32, >= n_flat + 1 elements: the flat gradient of this rank's LOCAL mean
    loss followed by one spare element).  weight_by_windows: the gradient is multiplied by this rank's window count, the count rides in the spare
    element, ONE sum-all-reduce moves both, and the returned divisor (a 0-dim tensor on buf's device, no host sync) is the global window count --
    buf[:n_flat] / divisor is then the gradient of the GLOBAL mean loss for ragged shards too (what `allreduce_gradients_` computes with sizes known
    on the host).  Otherwise: plain sum, divisor = world size (torch DDP's mean of the ranks' means, gnnLightning.py:1396-1400)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if live is not None and len(live) > 2 and not live[2].get("checked"):
        # first exchange: the live-element route is exact only if the engine really wrote zeros everywhere else (the Python liveness mirror and the
        # plan compiler agree, MSHGNN_PRUNE was the same when both were read) -- one host sync, once; otherwise fall back to the full exchange
        live[2]["checked"] = True
        dead = torch.ones(n_flat, dtype=torch.bool, device=buf.device)
        dead[live[0][live[0] < n_flat]] = False
        # (NaN counts as non-zero; the verdict is combined over the ranks -- a MAX all-reduce of one flag -- so that every rank takes the same route: ranks that
        #  disagreed would issue all-reduces of different element counts, which is a hang, not a fallback)
        bad_here = bool(dead.any()) and not bool((buf[:n_flat][dead] == 0).all())
        flag = torch.tensor([1.0 if bad_here else 0.0], dtype=torch.float32, device=buf.device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if float(flag) != 0.0:
            import warnings
            warnings.warn("flat_data_parallel(live_only=True): the engine's gradient is not zero outside spec.live_gradient_index() on some rank "
                          "(liveness mirror and plan disagree) -- exchanging the full buffer instead")
            live[2]["disabled"] = True
    if live is not None and len(live) > 2 and live[2].get("disabled"):
        live = None
    if live is not None:      # (flat_data_parallel(live_only=True)) the same exchange on the packed live elements [+ the count]: exact, the dead elements are zeros on every rank
        idx, packed = live[:2]
        torch.index_select(buf, 0, idx, out=packed)
        if weight_by_windows:
            packed.mul_(float(local_windows))
            packed[-1] = float(local_windows)      # (idx ends with n_flat: the spare element)
        _sum_all_reduce_(packed, group)
        buf.index_copy_(0, idx, packed)
        return packed[-1].clone() if weight_by_windows else torch.tensor(float(world), dtype=torch.float32, device=buf.device)
    if not weight_by_windows:
        _sum_all_reduce_(buf[:n_flat], group)
        return torch.tensor(float(world), dtype=torch.float32, device=buf.device)
    if buf.numel() < n_flat + 1:
        raise ValueError("exchange_flat_gradient_: the buffer needs one spare element behind the gradient")
    buf[:n_flat].mul_(float(local_windows))
    buf[n_flat:n_flat + 1].fill_(float(local_windows))
    _sum_all_reduce_(buf[:n_flat + 1], group)
    return buf[n_flat].clone()


def flat_data_parallel(module, group=None, weight_by_windows: bool = False, live_only: bool = False, stream_collective: bool = False):
    """Data parallelism for a training-step wrapper (wrappers.py) or a model (models.py) WITHOUT torch's DistributedDataParallel: the
    parameters (views of one flat buffer) are broadcast from rank 0 once, and from then on the fused training step
    (`training_step` -> `models.fused_training_step[_windows]`) sum-all-reduces its flat gradient in `loss.backward()` and divides by the
    world size -- the mean over the ranks' mean-loss gradients that DDP / Lightning-DDP produce (gnnLightning.py:1396-1400; the default, so a
    multi-GPU run of the reference and one through this function exchange the same gradient, ragged last batch included), as ONE exchange of 4 MB
    instead of ~50 per-parameter buckets and autograd hooks.  Every rank feeds its own shard of the global batch.  `weight_by_windows=True` is the
    opt-in deviation from the reference: each rank's gradient is weighted by its window count and the sum divided by the global count
    (`exchange_flat_gradient_`: the counts ride in the same all-reduce), so ragged shards -- the last batch of an epoch with drop_last=False --
    give the gradient of the GLOBAL mean loss instead of the mean of the ranks' means.  The two-call route (`forward` + `loss.backward()`: `fused_training_step = False`, a frozen parameter, a
    normalising WindowBatch) makes the same single exchange in the engine's backward.  Only the fused engine has that hook: a model that runs
    operator by operator (an activation other than ReLU) is rejected here -- wrap that one in
    torch's DistributedDataParallel.  The model must have seen its lazy-initialising forward and live on the device; do not also wrap it
    in DDP.  live_only: the exchange moves only the elements that can be non-zero at the model's depth (exact; A1-C2 at 3 layers: 36 % of the
    buffer -- `LiveGradientExchange`).  stream_collective: the exchange is enqueued on the step's own stream through the C-ABI (`StreamAllReduce`, RCCL bound at
    run time; GPUs only, one rank per device) instead of torch.distributed's all_reduce -- same sums, no hand-over to its side stream.  Returns `module`."""
    import torch.distributed as dist
    model = getattr(module, "model", module)
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("flat_data_parallel needs an initialised torch.distributed process group")
    if getattr(model, "_spec", None) is None:
        raise RuntimeError("flat_data_parallel: run the lazy-initialising forward first (the wrappers' constructors do)")
    if not getattr(model, "_fused_activation", True):
        raise RuntimeError("flat_data_parallel: this model runs operator by operator (an activation other than ReLU): its gradients never pass the engine's "
                           "flat buffer -- wrap it in torch.nn.parallel.DistributedDataParallel")
    p0 = model._params_in_flat_order()[0]
    if p0.device.type != "cuda":
        raise RuntimeError("flat_data_parallel: move the model to its GPU first")
    flat = model._flat_params(p0.device)
    dist.broadcast(flat, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if stream_collective and group not in _STREAM_COMMS:
        _STREAM_COMMS[group] = StreamAllReduce(flat.device, group)      # (collective: every rank of the group gets here)
    model._flat_ddp = True if group is None else group
    model._flat_ddp_weighted = bool(weight_by_windows)
    model._flat_ddp_live = None
    if live_only:      # exchange only the elements whose gradient can be non-zero at this depth (LiveGradientExchange's index), plus the window count's spare element
        idx = model._spec.live_gradient_index().to(flat.device)
        if weight_by_windows:
            idx = torch.cat([idx, torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)])
        model._flat_ddp_live = (idx, torch.empty(idx.numel(), dtype=torch.float32, device=flat.device), {"checked": False})
    return module
