"""CPU tests of the N>1 path (gloo, world_size 2): contiguous window sharding + one all-reduce of the flat
gradient reproduces the single-process gradient of the global batch.  The per-rank compute here is the
oracle (allowed in tests); on GPUs the same flat buffer comes out of mshgnn_backward and RCCL replaces gloo."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from morphsym_hgnn_amd import ddp, synth
from morphsym_hgnn_amd.engine import flatten_params
from tests import helpers


def test_shard_bounds_cover_the_batch():
    for B in (1, 7, 8, 8192, 8195):
        for world in (1, 2, 3, 8):
            spans = [ddp.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def _worker(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ms_hgnn_oracle as orc
    torch.set_num_threads(2)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    cfg = helpers.oracle_config(spec)
    x_dict, y = synth.make_windows(21, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(21, spec.param_shapes())
    xs, (b, e) = ddp.shard_x_dict(x_dict, spec.num_nodes, B, rank, world)
    n = e - b
    _, _, grads = orc.step(cfg, params, xs, spec.topology.edge_index_dict(n), y[b:e], n)
    flat = flatten_params(spec, grads).double()
    ddp.allreduce_gradients_(flat, n, B)
    if rank == 0:
        _, _, full = orc.step(cfg, params, x_dict, spec.topology.edge_index_dict(B), y, B)
        ref = flatten_params(spec, full).double()
        ret["err"] = float((flat - ref).abs().max() / ref.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_global_batch():
    world, B = 2, 5   # ragged split: 3 + 2 windows
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    assert ret["err"] < 1e-6   # fp32 flat buffer round-trip of fp64 oracle gradients


def _exchange_worker(rank, world, port, B, ret):
    """flat_data_parallel's exchange on a ragged split (ddp.exchange_flat_gradient_): every rank holds the flat gradient of its LOCAL mean loss in a
    buffer with one spare element; ONE all-reduce carries gradients and window counts; gradient / divisor == the global batch's gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ms_hgnn_oracle as orc
    torch.set_num_threads(2)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    cfg = helpers.oracle_config(spec)
    x_dict, y = synth.make_windows(23, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(23, spec.param_shapes())
    xs, (b, e) = ddp.shard_x_dict(x_dict, spec.num_nodes, B, rank, world)
    n = e - b
    _, _, grads = orc.step(cfg, params, xs, spec.topology.edge_index_dict(n), y[b:e], n)
    n_flat = spec.flat_size()
    for weighted in (True, False):
        buf = torch.zeros(n_flat + 16, dtype=torch.float32)
        buf[:n_flat] = flatten_params(spec, grads)
        div = ddp.exchange_flat_gradient_(buf, n_flat, n, None, weighted)
        got = (buf[:n_flat] / div).double()
        if weighted:
            assert float(div) == B
            if rank == 0:
                _, _, full = orc.step(cfg, params, x_dict, spec.topology.edge_index_dict(B), y, B)
                ref = flatten_params(spec, full).double()
                ret["err"] = float((got - ref).abs().max() / ref.abs().max())
        else:
            assert float(div) == world      # torch DDP's mean over the ranks' means
            mine = flatten_params(spec, grads).double()
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            ret[f"mean_err_{rank}"] = float((got - sum(both) / world).abs().max() / got.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_exchange_weights_ragged_shards_by_their_windows():
    world, B = 2, 7   # ragged split: 4 + 3 windows
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31000 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(world, port, B, ret), nprocs=world, join=True)
    assert ret["err"] < 1e-6 and ret["mean_err_0"] < 1e-6 and ret["mean_err_1"] < 1e-6


def _live_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    ex = ddp.LiveGradientExchange(spec, "cpu")
    g = torch.Generator().manual_seed(200 + rank)
    mine = torch.zeros(spec.flat_size())
    mine[ex.index] = torch.randn(ex.index.numel(), generator=g)      # a rank's gradient: exact zeros on the dead elements
    full = mine.clone()
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    full /= world
    got = ex.allreduce_mean_(mine.clone())
    ret[f"equal_{rank}"] = bool(torch.equal(got, full))
    ret["fraction"] = ex.fraction
    dist.barrier()
    dist.destroy_process_group()


def test_live_gradient_exchange_equals_the_full_exchange():
    """Exchanging only the elements that can be non-zero (A1-C2 at 3 layers: 35.6 % of the flat gradient) gives the full exchange's values, bit for bit."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 32500 + (os.getpid() % 2000)
    mp.spawn(_live_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["equal_0"] and ret["equal_1"] and 0.35 < ret["fraction"] < 0.36


def _bf16_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(4099, generator=g)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    got = ddp.allreduce_mean_bf16_(mine.clone(), torch.empty(4099, dtype=torch.bfloat16))
    ref = sum(both) / world
    ret[f"err_{rank}"] = float((got - ref).abs().max() / ref.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_exchange_is_within_bf16_distance_of_the_fp32_mean():
    """The opt-in bf16 gradient exchange (bench.py --grad-exchange bf16): mean over two ranks within 2^-7 of the fp32 mean, max-abs relative
    (three bf16 roundings) -- far outside the 1e-4 parity tolerance, which is why it is opt-in."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 32000 + (os.getpid() % 2000)
    mp.spawn(_bf16_worker, args=(world, port, ret), nprocs=world, join=True)
    assert 1e-4 < ret["err_0"] < 2 ** -7 and ret["err_0"] == ret["err_1"]


def _gpu_worker(rank, world, port, B, dtype, ret):
    """One rank of a data-parallel step on the REAL engine: both ranks share cuda:0 (the only GPU of the test box), so the exchange goes
    through gloo on host copies -- everything else (sharding, per-rank engine step through the C-ABI, mean semantics) is what bench.py does
    with RCCL on N GPUs."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from morphsym_hgnn_amd import engine as eng
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    x_dict, y = synth.make_windows(31, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(31, spec.param_shapes())
    e = eng.Engine(spec, dtype, device="cuda:0")
    flat = eng.flatten_params(spec, params, e.device)
    xs, (b, en) = ddp.shard_x_dict(x_dict, spec.num_nodes, B, rank, world)
    n = en - b
    _, _, g = e.step_mse(e.cast_inputs(xs), flat, y[b:en].reshape(-1).to(e.device, torch.float32), n)
    g = g.cpu()
    ddp.allreduce_gradients_(g, n, B)          # local-mean gradients -> gradient of the global mean loss
    if rank == 0:
        _, _, g_full = e.step_mse(e.cast_inputs(x_dict), flat, y.reshape(-1).to(e.device, torch.float32), B)
        ret["err"] = float((g - g_full.cpu()).abs().max() / g_full.abs().max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "x3"])
def test_two_engine_ranks_reproduce_the_global_batch_gradient(dtype):
    world, B = 2, 333   # ragged split: 167 + 166 windows
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31000 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(world, port, B, dtype, ret), nprocs=world, join=True)
    assert ret["err"] < 1e-5    # same arithmetic, different summation order over the windows


def _wrapper_worker(rank, world, port, B, live_only, weighted, ret, stream=False):
    """Two wrapper ranks on the one GPU of the test box (gloo: its all-reduce takes device tensors through host copies): flat_data_parallel +
    training_step + backward on each rank's shard == the single-process step on the whole batch."""
    import types
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from morphsym_hgnn_amd import wrappers
    torch.set_default_dtype(torch.float64)
    dev = torch.device("cuda", 0)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    _, cfg = helpers.load_group("a1-c2")
    x_dict, y = synth.make_windows(41, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(41 + rank, spec.param_shapes())          # different on every rank: the broadcast must make them rank 0's

    def batch_of(xd, yy, n):
        return types.SimpleNamespace(x_dict={k: v.to(dev) for k, v in xd.items()}, edge_index_dict=spec.topology.edge_index_dict(n, device=dev),
                                     y=yy.to(dev).flatten(), batch_size=n)
    xs, (b, en) = ddp.shard_x_dict(x_dict, spec.num_nodes, B, rank, world)
    mine = batch_of(xs, y[b:en], en - b)
    w = wrappers.HGNN_C2_Lightning_Reg(128, 2, spec.topology.metadata(), mine, symmetry_mode="MorphSym", group_operator_path=cfg).to(dev)
    w.model.load_state_dict(params)
    with torch.no_grad():
        w.model(x_dict=dict(mine.x_dict), edge_index_dict=mine.edge_index_dict)      # parameters become views of the flat buffer
    ddp.flat_data_parallel(w, live_only=live_only, stream_collective=stream, **({"weight_by_windows": True} if weighted else {}))      # (the default must be DDP's mean of means)
    assert (None in ddp._STREAM_COMMS) == stream
    assert w.model._flat_ddp_weighted == weighted
    assert (w.model._flat_ddp_live is not None) == live_only
    loss = w.training_step(mine, 0)
    assert w.model._gpend_id == 1                                      # the one-call step ran under torch.distributed
    w.model.zero_grad()
    loss.backward()
    g = w.model._gflat.clone()
    # the two-call route (forward + loss.backward()) under flat_data_parallel makes the same exchange in the engine's backward
    w.fused_training_step = False
    w.model.zero_grad()
    l2 = w.training_step(mine, 0)
    assert w.model._gpend_id == 1                                      # (no second one-call step)
    l2.backward()
    g2 = torch.zeros_like(g)
    for (o, n), q in zip(spec.param_offsets().values(), w.model._params_in_flat_order()):
        g2[o:o + n] = q.grad.flatten()
    ret[f"two_call_{rank}"] = float((g2 - g).abs().max() / g.abs().max())
    if rank == 0:
        ref = wrappers.HGNN_C2_Lightning_Reg(128, 2, spec.topology.metadata(), mine, symmetry_mode="MorphSym", group_operator_path=cfg).to(dev)
        ref.model.load_state_dict(synth.make_params(41, spec.param_shapes()))
        ref.model._flat_ddp = None
        ref.fused_training_step = False                                # (single-process reference through autograd)

        def grad_of(xd, yy, n):
            ref.model.zero_grad()
            ref.training_step(batch_of(xd, yy, n), 0).backward()
            gg = torch.zeros_like(g)                                   # (under torch.distributed the two-call route goes through autograd)
            for (o, m), q in zip(spec.param_offsets().values(), ref.model._params_in_flat_order()):
                gg[o:o + m] = q.grad.flatten()
            return gg
        if weighted:      # opt-in: the gradient of the GLOBAL mean loss, ragged shards included
            gf = grad_of(x_dict, y, B)
        else:             # the default, torch DDP / Lightning-DDP (gnnLightning.py:1396-1400): the mean over the ranks of each rank's mean-loss gradient
            gf = torch.zeros_like(g)
            for r in range(world):
                xr, (rb, re_) = ddp.shard_x_dict(x_dict, spec.num_nodes, B, r, world)
                gf += grad_of(xr, y[rb:re_], re_ - rb) / world
        ret["err"] = float((g - gf).abs().max() / gf.abs().max())
        ret["params_equal"] = bool(torch.equal(w.model._flat, ref.model._flat))
    dist.barrier()
    if stream:
        ddp._STREAM_COMMS.pop(None).close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B,live_only,weighted", [(64, False, False), (65, False, False), (65, False, True), (65, True, True), (65, True, False)])
def test_flat_data_parallel_wrappers_reproduce_the_global_batch_gradient(B, live_only, weighted):
    """Equal shards; a ragged split (33 + 32 windows) with the default exchange == DDP's mean of the ranks' means, and with weight_by_windows=True == the
    global-batch gradient; both again with only the live elements on the wire."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33000 + (os.getpid() % 2000)
    mp.spawn(_wrapper_worker, args=(world, port, B, live_only, weighted, ret), nprocs=world, join=True)
    assert ret["params_equal"] and ret["err"] < 1e-5
    assert ret["two_call_0"] < 1e-5 and ret["two_call_1"] < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,layers,hidden,regression", [("bf16", 3, 128, True), ("x3", 3, 128, True), ("bf16", 2, 128, True), ("f32", 3, 128, True),
                                                            ("bf16", 3, 128, False), ("x3", 2, 128, False), ("bf16", 3, 512, True), ("x3", 2, 256, True),
                                                            ("bf16", 5, 128, True)])
def test_engine_gradient_is_exactly_zero_outside_the_live_elements(dtype, layers, hidden, regression):
    """What makes the live-element exchange exact on the real engine: every element of the flat gradient that `spec.live_gradient_index` leaves out is an exact
    zero after a step (A1-C2 below the graph's diameter: the base nodes cannot reach the feet), and the live part is not all zero -- on the LDS-resident
    kernels (bf16 / split / fp32 plans), the cross-entropy step, the generic-width engine (hidden 256 / 512: FIN_ZERO ops of mshgnn_gen_plan.hpp) and at a
    depth where the base nodes and base_transform are live in the first layers (5 layers: the last layers still have dead relations)."""
    from morphsym_hgnn_amd import engine as eng
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", hidden, layers, regression=regression)
    B = 50
    x_dict, y = synth.make_windows(7, B, spec.num_nodes, spec.widths, 12 if regression else 4, classification=not regression)
    e = eng.Engine(spec, dtype)
    assert e.generic == (hidden != 128)
    flat = eng.flatten_params(spec, synth.make_params(7, spec.param_shapes()), e.device)
    if regression:
        out, loss, g = e.step_mse(e.cast_inputs(x_dict), flat, y.reshape(-1).to(e.device, torch.float32), B)
    else:
        out, loss, g = e.step_ce(e.cast_inputs(x_dict), flat, y.reshape(-1).to(e.device, torch.int32), B)
    ex = ddp.LiveGradientExchange(spec, e.device)
    dead = torch.ones(g.numel(), dtype=torch.bool, device=g.device)
    dead[ex.index] = False
    assert 0.0 < ex.fraction <= 1.0 and int(dead.sum()) == g.numel() - ex.index.numel()
    if bool(dead.any()):
        assert float(g[dead].abs().max()) == 0.0
    assert float(g[ex.index].abs().max()) > 0.0
    before = g.clone()
    ex.allreduce_mean_(g)                      # no process group: pack + scatter only
    assert torch.equal(before, g)


def _stream_comm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # (any backend: it only carries the 128-byte communicator id)
    from morphsym_hgnn_amd import engine as eng
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = ddp.StreamAllReduce(dev)
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    B = 64
    e = eng.Engine(spec, "bf16")
    x_dict, y = synth.make_windows(3, B, spec.num_nodes, spec.widths, 12)
    xs, yd = e.cast_inputs(x_dict), y.reshape(-1).to(dev, torch.float32)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), dev)
    out, loss, g = e.step_mse(xs, flat, yd, B)
    ref = g.clone()
    comm.allreduce_mean_(g)                                # one rank: the mean over the ranks is the buffer itself
    torch.cuda.synchronize()
    ret["same"] = bool(torch.equal(g, ref))
    # step + exchange as ONE captured HIP graph on a side stream (the collective is stream-ordered behind the step's last kernel)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        e.step_mse(xs, flat, yd, B, out=out, grad_flat=g, loss=loss); comm.allreduce_mean_(g)      # warm-up on the capture stream
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        g.zero_()
        with torch.cuda.graph(graph, stream=s):
            e.step_mse(xs, flat, yd, B, out=out, grad_flat=g, loss=loss)
            comm.allreduce_mean_(g)
        g.zero_()
        graph.replay()
    torch.cuda.synchronize()
    ret["graph_same"] = bool(torch.equal(g, ref))
    live = ddp.LiveGradientExchange(spec, dev)
    g2 = ref.clone()
    live.allreduce_mean_(g2, comm=comm)
    torch.cuda.synchronize()
    ret["live_same"] = bool(torch.equal(g2, ref))
    comm.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_stream_allreduce_through_the_c_abi_on_a_one_rank_group_and_inside_a_graph():
    """ddp.StreamAllReduce (mshgnn_comm_*: ncclAllReduce(avg) enqueued on the step's own stream, RCCL bound with dlopen): on a 1-rank communicator the mean
    is the identity -- bit for bit -- eagerly, captured in one HIP graph together with the step, and on the packed live elements."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stream_comm_worker, args=(1, 35000 + (os.getpid() % 2000), ret), nprocs=1, join=True)
    assert ret["same"] and ret["graph_same"] and ret["live_same"]


@pytest.mark.gpu
@pytest.mark.parametrize("live_only,weighted", [(False, False), (True, True)])
def test_flat_data_parallel_through_the_stream_collective_on_a_one_rank_group(live_only, weighted):
    """flat_data_parallel(stream_collective=True): the wrapper's one exchange goes through the C-ABI communicator on the step's stream (RCCL refuses two ranks on
    one GPU, so a 1-rank group: the exchange must leave the single-process gradient untouched on both routes, whole buffer and live elements + window count)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 36000 + (os.getpid() % 2000)
    mp.spawn(_wrapper_worker, args=(1, port, 64, live_only, weighted, ret, True), nprocs=1, join=True)
    assert ret["params_equal"] and ret["err"] < 1e-5 and ret["two_call_0"] < 1e-5
