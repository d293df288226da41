#!/usr/bin/env python3
"""bench.py -- graph-windows/sec, fwd + loss + bwd of the MS-HGNN hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config a1c2|mck4|solo|synth32] [--dtype bf16|x3|f32] [--surface flat|module]

A "step" is one pass of the hot path over one minibatch of synthetic windows that is already resident in HBM:
forward (encoder, L message-passing layers, decoder) + wrapper loss + backward with every parameter gradient
materialised (mshgnn_step_mse / mshgnn_forward + mshgnn_backward_ce), + the RCCL gradient all-reduce (mean over
ranks) when N > 1.  Prints ONE JSON line on rank 0.

Default workload = BASELINE.json configs[1]: A1 C2 GRF regression, bf16, h=128, L=3, grf_dimension=3, 8192
time-windows per GPU (weak scaling).  `--gpus N` with WORLD_SIZE unset starts the N rank processes itself (fresh
children, before this process makes any GPU call); under torch.distributed.run it uses the ranks it is given.

Timing: W warm-up steps, then blocks of EXACTLY K steps bracketed by barrier + synchronize on both sides, MAX over
ranks per block; blocks are repeated until >= 0.5 s have been timed and the MEDIAN block is reported (`timing` holds
the spread).  `roofline` is computed for the kernel with the largest share of the step from HIP events recorded
around every kernel by the C-ABI (mshgnn_profile_*) in a separate pass; `roofline.step` prices the WHOLE step against
SURVEY.md 8(d)'s algorithmic bytes / FLOPs.  `parity_plan` is the same workload on the parity-grade plan (<= 1e-4 of
the fp64 oracle, checked here on a small batch).  `cpu_baseline` times the fp64 oracle (a port of the reference's CPU
path, oracle/ms_hgnn_oracle.py) on the host cores on bounded samples (rank 0, N = 1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {  # /opt/skills/guides/MI355X_MICROARCH.md: chip-level parameters
    "hbm_GBs": 8000.0,
    "mfma_TFLOPs": {"f32": 157.3, "bf16": 2500.0, "x3": 2500.0},
}
# algorithmic minimum per A1-C2 window (h=128, T=150, d=3): inputs read once forward and once for the
# encoder's weight gradients, at the storage precision of the plan; FLOPs with dead nodes counted (the survey's figure)
SURVEY_8D = {"bytes_in_bf16": 14408.0, "flops_L3": 19.0e6, "flops_L8": 44.6e6}
PARITY_DTYPE = "x3"           # parity-grade plan reported next to the headline (falls back to f32 where x3 cannot run)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="a1c2", choices=["a1c2", "mck4", "solo", "synth32", "mcc2", "solo_s4", "mi_quad"])
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU (default: 8192 a1c2, 8192 mck4, 65536 solo, 1024 synth32)")
    ap.add_argument("--layers", type=int, default=0, help="message-passing layers (default: 3 a1c2, 8 mck4, 8 solo, 6 synth32)")
    ap.add_argument("--hidden", type=int, default=0, help="hidden channels (default 128; 512 for synth32)")
    ap.add_argument("--dtype", default=os.environ.get("MSHGNN_BENCH_DTYPE", "bf16"), choices=["f32", "bf16", "x3"])
    ap.add_argument("--surface", default="flat", choices=["flat", "module"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity-plan / L=8 / module-surface side measurements")
    ap.add_argument("--cpu-batch", type=int, default=1024)
    ap.add_argument("--min-time", type=float, default=0.5, help="seconds of timed blocks to accumulate")
    ap.add_argument("--overlap", default=os.environ.get("MSHGNN_BENCH_OVERLAP", "auto"), choices=["auto", "0", "1"])
    ap.add_argument("--grad-exchange", default=os.environ.get("MSHGNN_BENCH_GRAD_EXCHANGE", "auto"), choices=["auto", "f32", "live", "bf16"],
                    help="N > 1: what the gradient all-reduce moves.  f32: the whole flat buffer; live: only the elements that can be non-zero at this depth, "
                         "packed (exact -- ddp.LiveGradientExchange); auto (default): times f32 and live on this machine and keeps the faster; "
                         "bf16: opt-in, half the bytes, NOT parity-grade (ddp.allreduce_mean_bf16_)")
    ap.add_argument("--collective", default=os.environ.get("MSHGNN_BENCH_COLLECTIVE", "auto"), choices=["auto", "torch", "stream"],
                    help="N > 1: who enqueues the gradient all-reduce.  stream: ncclAllReduce on the step's own HIP stream through the C-ABI "
                         "(mshgnn_comm_allreduce_mean, ddp.StreamAllReduce); torch: torch.distributed.all_reduce (its side stream + event hand-over); "
                         "auto (default): times both on this machine and keeps the faster")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks (never from a process that has touched the GPU)
# ------------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv, env_extra=None, script=None, timeout_s: float = 3000.0) -> int:
    """Start n fresh rank processes of `script` (this file) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and watch ALL of them: rank 0's
    stdout is relayed (the JSON line), every rank's stderr goes to this process's stderr (ranks >= 1 also send their stdout there), and as soon
    as one rank exits non-zero -- or the time limit passes -- the others are terminated (a rank that died would leave its peers waiting in an
    RCCL collective until the collective's own timeout).  Children only: nothing here re-executes a process that has touched the GPU.
    Returns 0 when every rank exited 0, else the first non-zero exit code (124 on the time limit)."""
    import threading
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, t0 = 0, time.monotonic()
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.monotonic() - t0 > timeout_s:
            rc = bad[0] if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.monotonic()
            while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10.0:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            sys.stderr.write(f"bench.py: rank exit codes {codes} -> terminated the remaining ranks, exit {rc}\n")
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode())
        sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def build_spec(layers=3, config="a1c2", hidden=128):
    import yaml
    from morphsym_hgnn_amd import synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    cfgdir = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg")
    if config == "a1c2":
        with open(os.path.join(cfgdir, "a1-c2.yaml")) as f:
            group = yaml.safe_load(f)
        return ModelSpec(kind="c2", topology=topology.a1_c2(), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("c2", True), regression=True, grf_dimension=3, group=group)
    if config == "mck4":      # BASELINE configs[2]: MiniCheetah K4 contact-state classification (train_classification_msgn.py)
        with open(os.path.join(cfgdir, "mini_cheetah-k4.yaml")) as f:
            group = yaml.safe_load(f)
        return ModelSpec(kind="k4", topology=topology.TOPOLOGIES["mini_cheetah-k4"](), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("k4", False), regression=False, grf_dimension=3, group=group)
    if config == "solo":      # BASELINE configs[3]: Solo-12 K4 regression = the centroidal-momentum model COM_HGNN_K4 (train_regression-com_msgn.py:
        with open(os.path.join(cfgdir, "solo-k4.yaml")) as f:      # history_length 1, num_layers 8, hidden 128)
            group = yaml.safe_load(f)
        return ModelSpec(kind="k4_com", topology=topology.TOPOLOGIES["solo-k4-com"](), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("k4_com", True), regression=True, grf_dimension=3, group=group)
    if config == "mcc2":      # train_classification_msgn.py's default model type (heterogeneous_gnn_c2) on MiniCheetah: a compile-time program exists for it (tools/gen_spec_tables.py)
        with open(os.path.join(cfgdir, "mini_cheetah-c2.yaml")) as f:
            group = yaml.safe_load(f)
        return ModelSpec(kind="c2", topology=topology.TOPOLOGIES["mini_cheetah-c2"](), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("c2", False), regression=False, grf_dimension=3, group=group)
    if config == "solo_s4":   # train_regression-com_msgn.py's default model type (heterogeneous_gnn_s4_com): Solo-12 S4 centroidal-momentum regression
        with open(os.path.join(cfgdir, "solo-k4.yaml")) as f:
            group = yaml.safe_load(f)
        return ModelSpec(kind="s4_com", topology=topology.TOPOLOGIES["solo-s4-com"](), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("s4_com", True), regression=True, grf_dimension=3, group=group)
    if config == "mi_quad":   # the MI-HGNN baseline (hgnn.py:GRF_HGNN) on a quadruped, contact classification at the reference's depth (train_classification.py:650-652: heterogeneous_gnn, 8 layers, 128)
        return ModelSpec(kind="mi", topology=topology.TOPOLOGIES["quadruped-mi"](), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("mi", False), regression=False, grf_dimension=3, group=None)
    if config == "synth32":   # BASELINE configs[4]: synthetic 32-limb robot, MI-HGNN model (hgnn.py:GRF_HGNN)
        return ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=hidden, num_layers=layers,
                         widths=synth.feature_widths("mi", True), regression=True, grf_dimension=3, group=None)
    raise ValueError(config)


def defaults(args):
    d = {"a1c2": (8192, 3, 128), "mck4": (8192, 8, 128), "solo": (65536, 8, 128), "synth32": (1024, 6, 512), "mcc2": (8192, 8, 128), "solo_s4": (65536, 8, 128), "mi_quad": (8192, 8, 128)}[args.config]
    return (args.batch or d[0], args.layers or d[1], args.hidden or d[2])


def oracle_cfg(spec):
    from oracle import ms_hgnn_oracle as orc
    return orc.OracleConfig(kind=spec.kind, num_layers=spec.num_layers, edge_types=spec.edge_types, regression=spec.regression,
                            grf_dimension=spec.grf_dimension, group=spec.group, num_timesteps=spec.num_timesteps)


def make_batch(spec, B, seed):
    """Synthetic windows in the reference's calling convention: one IMU window tiled to every base node,
    joints ~N(0,1), foot ones (regression) / ~N(0,1) (classification inputs), labels ~N(0,1) or {0,1}."""
    import torch
    g = torch.Generator().manual_seed(seed)
    nn_, w = spec.num_nodes, spec.widths
    imu = torch.randn(B, 1, w["base"], generator=g)
    x = {"base": imu.expand(B, nn_["base"], w["base"]).reshape(B * nn_["base"], w["base"]),
         "joint": torch.randn(B * nn_["joint"], w["joint"], generator=g)}
    if "foot" in nn_:
        x["foot"] = torch.ones(B * nn_["foot"], 1) if w["foot"] == 1 else torch.randn(B * nn_["foot"], w["foot"], generator=g)
    n_out = nn_[spec.out_type]
    if spec.regression:
        y = torch.randn(B * n_out * spec.out_channels, generator=g)
    else:
        y = (torch.rand(B, n_out, generator=g) > 0.5).to(torch.int32)
    return x, y


def cpu_baseline(spec, batch, budget_s=20.0, scan=True):
    """Oracle (port of the reference CPU path, fp64) timed on the host cores: fwd + loss + bwd.  The thread count
    is picked from a short scan (small-matrix torch code does not scale to every core of the host)."""
    import torch
    from morphsym_hgnn_amd import synth
    from oracle import ms_hgnn_oracle as orc
    cfg = oracle_cfg(spec)
    n_y = spec.out_channels * spec.num_nodes[spec.out_type] if spec.regression else spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(1, batch, spec.num_nodes, spec.widths, n_y, classification=not spec.regression)
    params = synth.make_params(1, spec.param_shapes())
    ei = spec.topology.edge_index_dict(batch)

    def one():
        t0 = time.perf_counter()
        orc.step(cfg, params, x_dict, ei, y, batch)
        return time.perf_counter() - t0

    default_threads = torch.get_num_threads()
    best = (float("inf"), default_threads)
    t_start = time.perf_counter()
    for n in sorted({8, 16, 32, 64, default_threads}) if scan else sorted({1, 2, 4, 8, 16}):
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
            "sample": f"oracle fp64 fwd+loss+bwd, {spec.kind} h{spec.hidden} L{spec.num_layers}, B={batch}, median of {len(times)} steps, "
                      f"{best[1]} threads (best of a scan; host has {os.cpu_count()} cpus)"}


class Workload:
    """One (spec, plan dtype, batch) on one device: resident inputs + the step closure."""

    def __init__(self, spec, dtype, B, device, seed, dist=None, overlap="auto", grad_exchange="f32", collective="torch"):
        import torch
        from morphsym_hgnn_amd import engine as eng, synth
        self.torch, self.dist, self.spec, self.B = torch, dist, spec, B
        self.e = eng.Engine(spec, dtype=dtype, device=device)
        # NB resident batches, stepped through in turn: a timed loop that re-steps ONE batch could keep its 118 MB of bf16 inputs in the 256 MB Infinity
        # Cache between steps; three distinct ones (354 MB at the headline size, more than the cache) cannot.  `resident_batches` in the line says how many
        # the number was taken over, `single_resident_batch` gives the one-batch figure beside it.
        nb = int(os.environ.get("MSHGNN_BENCH_BATCHES", "3"))
        in_bytes = sum(spec.num_nodes[t] * spec.widths[t] for t in spec.node_types) * B * (2 if dtype == "bf16" else 4)
        while nb > 1 and nb * in_bytes > 8e9:      # (stay well inside the card for the large side configurations)
            nb -= 1
        self.batches = []
        for k in range(max(1, nb)):
            x, y = make_batch(spec, B, seed + 7919 * k)
            self.batches.append((self.e.cast_inputs(x), y.to(device)))
        self.nb, self.rot = len(self.batches), 0
        self.xs, self.y = self.batches[0]
        self.flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), device)
        self.gflat = torch.empty_like(self.flat)
        self.out = torch.empty(B * spec.num_nodes[spec.out_type], spec.out_channels, dtype=torch.float32, device=device)
        self.loss = torch.empty(1, dtype=torch.float32, device=device)
        split = int(self.e.info.grad_split)
        world = dist.get_world_size() if dist is not None else 1
        # Two-phase step (--overlap 1; --overlap auto, the default, measures both on the machine it runs on when there is more than one rank and
        # keeps the faster: calibrate_overlap): the all-reduce of everything but the encoder's gradients (83 % of the buffer) runs under the
        # encoder's weight-gradient launch.  Measured on one GPU with a 1-rank RCCL group (no wire time) the split
        # costs +51 us per step (two weight-gradient launches that each fill the chip less well, two finalize launches, two stream
        # hand-offs; +126 us when each phase gets its own window-part count: more slabs to write and sum), while an 8-GPU all-reduce of
        # 3.3 MB is ~60-90 us of which at most the ~45 us of the encoder's weight gradients can be hidden -- the plain sequence
        # "step, then one mean all-reduce of the 4 MB flat gradient" is at least as fast at this gradient size.
        self.can_overlap = dist is not None and split > 0 and spec.regression
        self.overlap = self.can_overlap and overlap == "1"
        self.split = split
        self.overlap_choice = None
        self.g16 = torch.empty(self.gflat.numel(), dtype=torch.bfloat16, device=device) if (dist is not None and grad_exchange == "bf16") else None
        self.live = None
        if dist is not None and grad_exchange in ("auto", "live"):
            from morphsym_hgnn_amd import ddp
            self.live = ddp.LiveGradientExchange(spec, device)
            if grad_exchange == "auto" and self.live.fraction > 0.9:      # nothing worth packing
                self.live = None
        self.use_live = self.live is not None and grad_exchange == "live"
        self.exchange_choice = None
        # the whole-buffer fp32 exchange can be enqueued on the step's own stream through the C-ABI (ddp.StreamAllReduce) instead of torch.distributed
        self.stream_comm, self.use_stream, self.collective_choice = None, False, None
        if dist is not None and collective in ("auto", "stream"):
            from morphsym_hgnn_amd import ddp
            try:
                self.stream_comm = ddp.StreamAllReduce(device)
                self.use_stream = collective == "stream"
            except Exception as ex:  # noqa: BLE001  (no librccl the loader can find: torch.distributed's collective stays)
                sys.stderr.write(f"bench.py: StreamAllReduce unavailable ({ex}); using torch.distributed.all_reduce\n")
                self.collective_choice = {"mode": collective, "stream": False, "error": str(ex)[:200]}
            # every rank must take the same route (the calibration passes and the step itself are collectives): one rank without the communicator
            # puts all of them back on torch.distributed's all-reduce
            ok = torch.tensor([1.0 if self.stream_comm is not None else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok[0]) == 0.0 and self.stream_comm is not None:
                self.stream_comm.close()
                self.stream_comm, self.use_stream = None, False
                self.collective_choice = {"mode": collective, "stream": False, "error": "another rank could not create the communicator"}

    def _allreduce(self):
        if self.g16 is not None:
            from morphsym_hgnn_amd import ddp
            ddp.allreduce_mean_bf16_(self.gflat, self.g16)
        elif self.use_live:
            self.live.allreduce_mean_(self.gflat, comm=self.stream_comm if self.use_stream else None)
        elif self.use_stream:
            self.stream_comm.allreduce_mean_(self.gflat)      # ncclAllReduce(avg) on this stream, behind the step's last kernel
        else:
            self.dist.all_reduce(self.gflat, op=self.dist.ReduceOp.AVG)     # DDP semantics: mean over ranks (gnnLightning.py:1396-1400)

    def step(self):
        e, dist = self.e, self.dist
        if self.nb > 1:      # the next resident batch
            self.xs, self.y = self.batches[self.rot % self.nb]
            self.rot += 1
        if not self.spec.regression:      # classification wrapper: forward + cross entropy + backward in one call (mshgnn_step_ce)
            e.step_ce(self.xs, self.flat, self.y, self.B, out=self.out, grad_flat=self.gflat, loss=self.loss)
            if dist is not None:
                self._allreduce()
            return self.loss
        if not self.overlap:
            e.step_mse(self.xs, self.flat, self.y, self.B, out=self.out, grad_flat=self.gflat, loss=self.loss)
            if dist is not None:
                self._allreduce()
            return self.loss
        # N > 1: the same step in two calls; the all-reduce of everything but the encoder's gradients (83 % of the buffer) runs on
        # RCCL's stream while this stream computes the encoder's weight gradients, then the encoder's slice follows.  Both
        # collectives complete inside the step (wait() makes this stream wait for them).
        e.step_mse_phase(0, self.xs, self.flat, self.y, self.B, self.out, self.gflat, self.loss)
        w1 = dist.all_reduce(self.gflat[self.split:], op=dist.ReduceOp.AVG, async_op=True)
        e.step_mse_phase(1, self.xs, self.flat, self.y, self.B, self.out, self.gflat, self.loss)
        w2 = dist.all_reduce(self.gflat[:self.split], op=dist.ReduceOp.AVG, async_op=True)
        w1.wait(); w2.wait()
        return self.loss

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _time_mode(self, steps, warmup):
        torch, dist = self.torch, self.dist
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.e.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0]) / steps * 1e3

    def calibrate_collective(self, steps=20, warmup=3):
        """--collective auto with more than one rank: the plain sequence with torch.distributed's all-reduce and with the C-ABI's on the step's own stream,
        MAX over ranks, keep the faster."""
        live, ov = self.use_live, self.overlap
        self.use_live, self.overlap = False, False
        ms = {}
        for mode in (False, True):
            self.use_stream = mode
            ms[mode] = self._time_mode(steps, warmup)
        self.use_stream = ms[True] <= ms[False]
        self.use_live, self.overlap = live, ov
        self.collective_choice = {"mode": "auto", "stream": bool(self.use_stream), "ms_torch_distributed": ms[False], "ms_stream_c_abi": ms[True]}

    def calibrate_exchange(self, steps=20, warmup=3):
        """--grad-exchange auto with more than one rank: time the step with the whole-buffer all-reduce and with the packed live-element exchange
        (both exact) on THIS machine, MAX over ranks, and keep the faster."""
        ms = {}
        for mode in (False, True):
            self.use_live = mode
            ms[mode] = self._time_mode(steps, warmup)
        self.use_live = ms[True] < ms[False]
        self.exchange_choice = {"mode": "auto", "live": bool(self.use_live), "live_fraction": self.live.fraction,
                                "ms_whole_buffer": ms[False], "ms_live_packed": ms[True]}

    def calibrate_overlap(self, steps=20, warmup=3):
        """--overlap auto with more than one rank: time the plain sequence (step, then ONE all-reduce of the flat gradient) and the two-phase step
        (all-reduce of everything but the encoder's gradients under the encoder's weight-gradient launch) on THIS machine -- MAX over ranks, so every
        rank takes the same decision -- and keep the faster.  What an 8-GPU node's all-reduce costs is not something a 1-GPU box can tell."""
        torch, dist = self.torch, self.dist
        ms = {}
        for mode in (False, True):
            self.overlap = mode
            for _ in range(warmup):
                self.step()
            torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.e.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms[mode] = float(t[0]) / steps * 1e3
        self.overlap = ms[True] < ms[False]
        self.overlap_choice = {"mode": "auto", "two_phase": bool(self.overlap), "ms_plain": ms[False], "ms_two_phase": ms[True]}

    def time_blocks(self, steps, warmup, min_time, max_blocks=200):
        """-> (median seconds per K-step block, list of all block times), MAX over ranks per block."""
        torch, dist = self.torch, self.dist
        for _ in range(warmup):
            self.step()
        blocks, total = [], 0.0
        while len(blocks) < max_blocks and (not blocks or total < min_time):
            torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize(); self.barrier(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt, total + dt], dtype=torch.float64, device=self.e.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, total = float(t[0]), float(t[1])      # every rank sees the same times -> the same number of blocks
            else:
                total += dt
            blocks.append(dt)
        srt = sorted(blocks)
        return srt[len(srt) // 2], blocks

    def kernel_stats(self, steps):
        e = self.e
        e.profile(True)
        for _ in range(steps):
            self.step()
        self.torch.cuda.synchronize()
        stats = [s for s in e.profile_read() if s["launches"] > 0]
        e.profile(False)
        return stats


def median_step_s(step, sync, steps, warmup, min_time=0.25, max_blocks=50):
    """Seconds per call of `step()`: warm-up, then blocks of `steps` calls until >= min_time seconds have been timed; the MEDIAN block."""
    for i in range(warmup):
        step()
    blocks, total = [], 0.0
    while len(blocks) < max_blocks and (len(blocks) < 3 or total < min_time):
        sync(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync(); dt = time.perf_counter() - t0
        blocks.append(dt / steps); total += dt
    return sorted(blocks)[len(blocks) // 2]


def parity_golden(dtype, device, name="a1c2_h128_L3_d3_B3"):
    """The UNCONDITIONAL parity check of plan `dtype`: the committed golden vectors of the reference run (tests/golden/<name>.npz: output, loss and
    per-gradient norm / samples / sums with the reference's own relu decisions) against the engine's results, in check_against_fixture's terms,
    plus how many of the engine's relu decisions differ from the exact ones on that case (each must sit within 1e-4 of its tensor's scale of zero)."""
    from tests import helpers
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype=dtype, device=str(device))
    flipped = int(getattr(helpers.run_engine_case, "last_decisions_differing", 0))
    out_err, worst = helpers.check_against_fixture(fx, out, loss, grads, rtol=1e-4, what="bench parity_plan")
    return {"golden_case": name, "out_rel_err": out_err, "worst_gradient_term": worst[0], "worst_gradient": worst[1],
            "flipped_relu_decisions": flipped, "relu_decisions_outside_tolerance": int(errs["relu_decisions_outside_tolerance"]),
            "what": "engine vs the committed golden vectors of the reference run (exact relu decisions): output max-abs / max-abs, "
                    "per-gradient L2-norm / sample / sum terms of tests/helpers.check_against_fixture; tolerance 1e-4"}


def parity_error(spec, dtype, device, B=48):
    """Worst max-abs error / max-abs reference over every hidden state, the output, the loss and every parameter gradient of plan
    `dtype` against the fp64 oracle on B seeded windows -- the GPU tests' own harness (tests/helpers.run_engine_case: the oracle is
    evaluated with the engine's relu decisions, each of which must lie within 1e-4 of the exact one's zero crossing)."""
    from morphsym_hgnn_amd import synth
    from tests import helpers
    n_y = spec.out_channels * spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(77, B, spec.num_nodes, spec.widths, n_y)
    params = synth.make_params(77, spec.param_shapes())
    errs, *_ = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype=dtype, device=str(device))
    outside = errs.pop("relu_decisions_outside_tolerance")
    return max(errs.values()), int(getattr(helpers.run_engine_case, "last_decisions_differing", 0)), int(outside)


def module_surface(spec, B, device, steps, warmup, precision):
    """The nn.Module surface the reference's Lightning wrappers call (gnnLightning.py:680-722): x_dict resident on the device ->
    GRF_HGNN_C2.forward -> MSE -> loss.backward(), parameters as nn.Parameters.  Two input conventions: fp64 as the reference's datasets
    produce them (the encoder reads them as they are and converts in registers: engine.WideInputs, mshgnn_forward_src -- `cast_pass` is the
    same with MSHGNN_WIDE_SRC=0, i.e. a separate fp64 -> plan-dtype cast of 59 M elements in front of every step, rounds 1-4), and tensors already
    at the plan's input dtype and pitch (what the on-device window assembly hands over: no cast)."""
    import torch
    from morphsym_hgnn_amd import models, synth
    from morphsym_hgnn_amd.checkpoint import load_into
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)      # the reference does (gnnLightning.py:1183)
    res = {"precision": precision, "what": "GRF_HGNN_C2.forward(x_dict on device, edge_index_dict) + MSE + loss.backward(), nn.Parameter weights"}
    try:
        import types
        from morphsym_hgnn_amd import wrappers
        cfg = os.path.join(ROOT, "morphsym_hgnn_amd", "cfg", "a1-c2.yaml")
        x, y = make_batch(spec, B, 99)
        x64 = {k: v.to(device, torch.float64) for k, v in x.items()}
        y = y.to(device, torch.float64).view(B, -1)
        ei = spec.topology.edge_index_dict(B, device=device)
        prev_env = os.environ.get("MSHGNN_DTYPE")      # (the wrapper builds its model itself: the plan is chosen through MSHGNN_DTYPE)
        os.environ["MSHGNN_DTYPE"] = precision
        try:      # the reference's own entry: the wrapper builds the model and runs the lazy-initialising dummy forward (gnnLightning.py:564-595)
            w = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, spec.num_layers, spec.topology.metadata(),
                                               types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei), lr=1e-4, symmetry_mode="MorphSym",
                                               group_operator_path=cfg)
        finally:
            if prev_env is None:
                os.environ.pop("MSHGNN_DTYPE", None)
            else:
                os.environ["MSHGNN_DTYPE"] = prev_env
        m = w.model
        load_into(m, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
        m.set_precision(precision)
        w.to(device)
        with torch.no_grad():
            m(dict(x64), ei)
        e = next(iter(m._engines.values()))
        os.environ["MSHGNN_WIDE_SRC"] = "0"      # (the separate cast + re-pitch pass, once: these are the tensors on-device window assembly would hand over)
        try:
            xplan = dict(zip(e.types, e.cast_inputs(x64)))
        finally:
            del os.environ["MSHGNN_WIDE_SRC"]
        y32 = y.float()      # labels as the on-device window assembly hands them over

        def run(xin, device_loss=False):
            def step():
                m.zero_grad(set_to_none=True)
                if device_loss:      # the wrapper's own training_step: model + metric bookkeeping on the device, its mse_loss carries autograd
                    loss = w.training_step(types.SimpleNamespace(x_dict=dict(xin), edge_index_dict=ei, y=y32, batch_size=B), 0)
                else:
                    out = m(dict(xin), ei)
                    loss = ((out.flatten() - y.flatten()) ** 2).mean()
                loss.backward()
            # >= 0.25 s of blocks, median (as the headline).  Two passes per route, the FIRST discarded as warm-up whatever it reads (the first pass of a route
            # after another route ran has read up to 20 % high for its whole quarter second on some boxes: tools/module_surface_repeat.py); both are reported
            passes = [median_step_s(step, torch.cuda.synchronize, steps, warmup) for _ in range(2)]
            all_passes.append([p * 1e3 for p in passes])
            return passes[1]

        all_passes = []
        dt64, dtp, dtm = run(x64), run(xplan), run(xplan, True)
        dtm64 = run(x64, True)
        os.environ["MSHGNN_WIDE_SRC"] = "0"
        try:
            dtc = run(x64)
        finally:
            del os.environ["MSHGNN_WIDE_SRC"]
        res["passes_ms"] = {"rule": "second of two passes per route (the first is a discarded warm-up pass), routes in the order listed",
                            "fp64": all_passes[0], "plan_dtype_inputs": all_passes[1], "wrapper_training_step": all_passes[2],
                            "wrapper_training_step_fp64_inputs": all_passes[3], "cast_pass": all_passes[4]}
        # the reference's own batch size (32, train_regression-grf_msgn.py:93): a step is a handful of short launches and the Python between them dominates --
        # the whole step (zero_grad + training_step + backward + FlatAdam.step) eager, and captured once in a HIP graph and replayed (wrappers.GraphedTrainingStep)
        try:
            res["B32"] = small_batch_surface(spec, device, precision, cfg, 32)
        except Exception as ex:      # (reported, never fatal to the headline)
            res["B32"] = {"error": repr(ex)[:300]}
        res.update({"ms_per_step": dt64 * 1e3, "value": B / dt64, "inputs": "fp64 on device (the reference's convention): read by the encoder as they are (mshgnn_forward_src), no cast pass",
                    "plan_dtype_inputs": {"ms_per_step": dtp * 1e3, "value": B / dtp, "inputs": "already at the plan's input dtype and pitch (no cast)"},
                    "cast_pass": {"ms_per_step": dtc * 1e3, "value": B / dtc, "inputs": "fp64 on device through a separate cast + re-pitch pass (MSHGNN_WIDE_SRC=0)"},
                    "wrapper_training_step_fp64_inputs": {"ms_per_step": dtm64 * 1e3, "value": B / dtm64,
                                                          "what": "wrapper_training_step with the reference's fp64 device inputs (no cast pass)"},
                    "wrapper_training_step": {"ms_per_step": dtm * 1e3, "value": B / dtm,
                                                      "what": "wrappers.HGNN_C2_Lightning_Reg.training_step(batch) + loss.backward(): as plan_dtype_inputs with fp32 "
                                                              "labels; the loss, the step metrics (MSE / RMSE / L1 sums, epoch accumulation) and dL/dy_pred come "
                                                              "from one device launch instead of torch's fp64 elementwise loss kernels"}})
    finally:
        torch.set_default_dtype(prev)
    return res


def small_batch_surface(spec, device, precision, cfg, B):
    """wrapper.training_step + backward + optimizer step at a small batch, fp64 device inputs: eager vs one replayed HIP graph."""
    import types
    import torch
    from morphsym_hgnn_amd import synth, wrappers
    from morphsym_hgnn_amd.checkpoint import load_into
    x, y = make_batch(spec, B, 98)
    x64 = {k: v.to(device, torch.float64) for k, v in x.items()}
    ei = spec.topology.edge_index_dict(B, device=device)
    prev_env = os.environ.get("MSHGNN_DTYPE")
    os.environ["MSHGNN_DTYPE"] = precision
    try:
        w = wrappers.HGNN_C2_Lightning_Reg(spec.hidden, spec.num_layers, spec.topology.metadata(), types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei),
                                           lr=1e-4, symmetry_mode="MorphSym", group_operator_path=cfg)
    finally:
        if prev_env is None:
            os.environ.pop("MSHGNN_DTYPE", None)
        else:
            os.environ["MSHGNN_DTYPE"] = prev_env
    load_into(w.model, {"state_dict": {"model." + k: v for k, v in synth.make_params(0, spec.param_shapes()).items()}})
    w.model.set_precision(precision)
    w.to(device)
    w.graph_safe_optimizer = True
    opt = w.configure_optimizers()
    batch = types.SimpleNamespace(x_dict=dict(x64), edge_index_dict=ei, y=y.to(device, torch.float64).view(B, -1), batch_size=B)

    def eager():
        opt.zero_grad(set_to_none=True)
        loss = w.training_step(batch, 0)
        loss.backward()
        opt.step()
    t_eager = median_step_s(eager, torch.cuda.synchronize, 50, 10)
    gs = wrappers.GraphedTrainingStep(w, opt, batch)
    t_graph = median_step_s(lambda: gs(batch), torch.cuda.synchronize, 50, 10)      # (includes copying the batch into the graph's static tensors)
    return {"windows": B, "what": "zero_grad + wrapper.training_step(fp64 device batch) + backward + FlatAdam.step, bf16 plan",
            "eager_ms_per_step": t_eager * 1e3, "graphed_ms_per_step": t_graph * 1e3,
            "graphed": "wrappers.GraphedTrainingStep: the step captured once in a HIP graph, replayed per batch (batch copied into static tensors first)"}


def end_to_end(spec, B, device, steps, warmup, config="a1c2"):
    """Training throughput INCLUDING the data path (never `value`): random window starts -> gather from a resident synthetic sequence ->
    step -> Adam on the flat buffers, all on the device (examples/train_flat.py).  Two routes with identical bits: mshgnn_assemble_windows +
    mshgnn_step_mse / mshgnn_step_ce, and mshgnn_step_mse_series / mshgnn_step_ce_series (the gather fused into the encoder)."""
    import numpy as np
    import torch
    from morphsym_hgnn_amd import engine as eng, synth
    from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe, minicheetah_k4_recipe
    rows, T = 200_000, 150
    rng = np.random.default_rng(0)
    ce = config == "mck4"
    if ce:      # MiniCheetah contact data format: labels = the contact flags of the window's last step
        seq = {k: rng.standard_normal((rows, c)).astype(np.float32) for k, c in (("imu_acc", 3), ("imu_omega", 3), ("q", 12), ("qd", 12), ("p", 12), ("v", 12))}
        seq["contacts"] = (rng.random((rows, 4)) < 0.5).astype(np.float32)
        recipe = minicheetah_k4_recipe(range(12), range(4), T)
    else:
        seq = {k: rng.standard_normal((rows, c)).astype(np.float32) for k, c in (("imu_acc", 3), ("imu_omega", 3), ("q", 12), ("qd", 12), ("tau", 12), ("F", 12), ("r_o", 4))}
        recipe = quadsdk_a1_c2_recipe(range(12), range(4), T, 3)
    res = {"what": f"random starts -> window gather from a resident {rows}-step sequence -> fwd + {'cross entropy' if ce else 'MSE'} + bwd -> Adam, "
                   f"{B} windows/step, bf16 plan (parity_plan: the same on the split-bf16 plan, fp32 series and windows)"}
    for plan in ("bf16", PARITY_DTYPE):
        res_p = res if plan == "bf16" else res.setdefault("parity_plan", {"dtype": plan})
        store = SequenceStore(seq, recipe, dtype=plan, device=device)
        e = eng.Engine(spec, dtype=plan, device=device)
        flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), device)
        gflat, m, v = torch.empty_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
        out = torch.empty(B * 4, spec.out_channels, dtype=torch.float32, device=device); loss = torch.empty(1, dtype=torch.float32, device=device)
        gen = torch.Generator(device=device).manual_seed(7)
        _end_to_end_routes(res_p, e, store, flat, gflat, m, v, out, loss, gen, B, ce, steps, warmup, device)
        del store, e
        torch.cuda.empty_cache()
    return res


def _end_to_end_routes(res, e, store, flat, gflat, m, v, out, loss, gen, B, ce, steps, warmup, device):
    import torch
    for name, fused in (("assemble_then_step", False), ("fused_gather", True)):
        def step(i, fused=fused):
            starts = torch.randint(0, len(store), (B,), generator=gen, device=device)
            if fused:
                (e.step_ce_series if ce else e.step_mse_series)(store, starts, flat, out=out, grad_flat=gflat, loss=loss)
            else:
                xs, y, _ = store.assemble(starts, reuse_buffers=True)
                if ce:
                    e.step_ce(xs, flat, (y != 0).to(torch.int32).view(B, -1), B, out=out, grad_flat=gflat, loss=loss)
                else:
                    e.step_mse(xs, flat, y.view(-1), B, out=out, grad_flat=gflat, loss=loss)
            e.adam_step(flat, gflat, m, v, i + 1, 1e-4)
        it = [0]

        def one():
            it[0] += 1
            step(it[0])
        dt = median_step_s(one, torch.cuda.synchronize, steps, warmup)      # >= 0.25 s of blocks, median (as the headline)
        res[name] = {"ms_per_step": dt * 1e3, "value": B / dt}


def source_hash():
    """sha1 over the product's sources (kernels, C-ABI, host package, this file): the same value in a bench line and in a committed profile means the
    profile was taken from exactly the code that produced the line, even when the profile was committed later (a commit cannot name its own hash)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "morphsym_hgnn_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
                   + glob.glob(os.path.join(ROOT, "morphsym_hgnn_amd", "*.py")) + [os.path.join(ROOT, "bench.py")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


WORKLOAD_TAGS = ("L8", "mck4", "solo", "synth32")


def workload_tag(config, B, L, hidden):
    """Which committed profile set (profiles/<round>_[<tag>_]<dtype>_pmc_traffic.json) belongs to a workload: "" = the headline (A1-C2 L=3, 8192 windows),
    "L8" = the paper's depth, "mck4" / "solo" / "synth32" = BASELINE configs[2..4] at their default sizes; None = no committed profile."""
    class A: pass
    a = A(); a.config, a.batch, a.layers, a.hidden = config, 0, 0, 0
    if (B, L, hidden) == defaults(a):
        return "" if config == "a1c2" else config
    if config == "a1c2" and (B, L, hidden) == (8192, 8, 128):
        return "L8"
    return None


def committed_traffic(tag, dtype, kernel):
    """(HBM bytes per launch of `kernel`, source file + commit) from the newest committed PMC traffic file of that workload and plan, or None."""
    if tag is None or dtype not in ("bf16", "x3"):
        return None
    try:
        import glob
        files = []
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{dtype}_pmc_traffic.json"))):
            name = os.path.basename(f)
            has = [t for t in WORKLOAD_TAGS if f"_{t}_" in name]
            if (has == [tag]) if tag else not has:
                files.append(f)
        if not files:
            return None
        meta = json.load(open(files[-1]))
        if kernel not in meta["kernels"]:
            return None
        src = os.path.basename(files[-1]) + (f" (commit {meta['commit']})" if "commit" in meta else "")
        if "source_hash" in meta:
            src += f" (source_hash {meta['source_hash']}: {'same sources as this line' if meta['source_hash'] == source_hash() else 'OTHER sources than this line'})"
        return meta["kernels"][kernel]["hbm_bytes"], src
    except Exception:  # noqa: BLE001
        return None


def roofline_of(stats, steps, B, dtype, e, config, L, hidden):
    """`roofline` of the dominant kernel of a step.  Top-level achieved / frac price the launch with SURVEY.md 8(d)'s ALGORITHMIC work only:
    HBM-bound kernels by the raw-input bytes they must read (+ the flat gradient written once for the weight-gradient kernel), MFMA-bound kernels
    by the plan's algorithmic FLOPs (dead nodes not counted).  What the kernel streams beyond that (stashed activations: operands of its own
    making) is reported under `operands`, never as the headline fraction."""
    total_ms = sum(s["total_ms"] for s in stats)
    dom = max(stats, key=lambda s: s["total_ms"])
    avg_s = dom["total_ms"] / dom["launches"] * 1e-3
    per_step = max(1, round(dom["launches"] / steps))      # the two-phase step launches the weight-gradient kernel twice
    flops_w, bytes_w = dom["flops_per_window"] / per_step, dom["bytes_per_window"] / per_step
    if dom["bound"] == "mfma":
        achieved = flops_w * B / avg_s / 1e12
        peak = PEAK["mfma_TFLOPs"][dtype]
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "algorithmic_flops": flops_w * B,
                "operands": {"bytes": bytes_w * B, "achieved": bytes_w * B / avg_s / 1e9, "unit": "GB/s", "frac": bytes_w * B / avg_s / 1e9 / PEAK["hbm_GBs"],
                             "note": "the launch's own operand stream against the HBM peak (tile in, stashes out): the other roof of this kernel"}}
    else:
        b8d = float(e.info.bytes_in) * B + (4.0 * e.spec.flat_size() if dom["name"].startswith("gradw") else 0.0)
        achieved = b8d / avg_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK["hbm_GBs"], "unit": "GB/s", "frac": achieved / PEAK["hbm_GBs"], "traffic": None,
                "algorithmic_bytes": b8d, "priced_by": "SURVEY 8(d) raw-input bytes",
                "operands": {"bytes": bytes_w * B, "achieved": bytes_w * B / avg_s / 1e9, "frac": bytes_w * B / avg_s / 1e9 / PEAK["hbm_GBs"],
                             "note": "every operand of the launch counted once, stashed activations included (the kernel's own operand floor)"}}
        if bytes_w * B > 20.0 * b8d:
            # a workload whose raw inputs are tiny next to its hidden state (Solo-12 COM: T = 1, 36 input values against 16 x 128 hidden per layer): the
            # raw-input figure says nothing about the launch (it read 0.0005 on that config) -- price it by its own operand stream, keep 8(d) beside it
            roof["survey_8d"] = {"algorithmic_bytes": b8d, "achieved": achieved, "frac": achieved / PEAK["hbm_GBs"]}
            roof["achieved"], roof["frac"], roof["algorithmic_bytes"] = roof["operands"]["achieved"], roof["operands"]["frac"], bytes_w * B
            roof["priced_by"] = "the launch's own operand stream (raw inputs are < 5 % of it)"
    # HBM-side traffic per launch of that kernel, from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    # WRITE_SIZE runs of this same command, gfx950 corrections applied -- tools/summarize_pmc.py); null if absent
    tr = committed_traffic(workload_tag(config, B, L, hidden), dtype, dom["name"].rstrip("0123456789"))
    if tr is not None:
        roof["traffic"] = tr[0] / per_step
        roof["traffic_source"] = tr[1]
    roof["kernel"] = dom["name"]
    roof["launches_per_step"] = per_step
    roof["avg_us"] = avg_s * 1e6
    roof["share_of_step"] = dom["total_ms"] / max(total_ms, 1e-9)
    # every launch of the step against its own roof, by the same algorithmic figures (the plan's live FLOPs / the raw-input bytes a launch must read): the
    # dominant launch changes with the build (round 5: the MFMA-classified stack launch; round 6: the HBM-classified weight-gradient launch), the table does not
    per = {}
    for s_ in stats:
        if not s_["launches"]:
            continue
        n_ = max(1, round(s_["launches"] / steps))
        t_ = s_["total_ms"] / s_["launches"] * 1e-3
        if s_["bound"] == "mfma":
            ach = s_["flops_per_window"] / n_ * B / t_ / 1e12
            per[s_["name"]] = {"bound": "mfma", "avg_us": t_ * 1e6, "achieved": ach, "unit": "TFLOP/s", "frac": ach / PEAK["mfma_TFLOPs"][dtype],
                               "algorithmic_flops": s_["flops_per_window"] / n_ * B}
        else:
            ob = s_["bytes_per_window"] / n_ * B
            per[s_["name"]] = {"bound": "hbm", "avg_us": t_ * 1e6, "achieved": ob / t_ / 1e9, "unit": "GB/s", "frac": ob / t_ / 1e9 / PEAK["hbm_GBs"],
                               "operand_bytes": ob, "priced_by": "the launch's own operand stream"}
    roof["per_kernel"] = per
    return roof


def side_config(config, device, steps, warmup, min_time=0.25):
    """One of the other BASELINE configs on this GPU, driver-visible inside the default line: throughput plan (bf16) and parity-grade plan, each the
    median of >= min_time seconds of blocks, + the dominant kernel's roofline fraction (as `roofline` of the headline)."""
    import torch
    class A: pass
    a = A(); a.config, a.batch, a.layers, a.hidden = config, 0, 0, 0
    B, L, hidden = defaults(a)
    spec = build_spec(L, config, hidden)
    names = {"mck4": "MiniCheetah-K4 contact classification (BASELINE configs[2] shape)", "solo": "Solo-12 K4 centroidal-momentum regression (configs[3])",
             "synth32": "synthetic 32-limb MI-HGNN, MFMA-bound stress (configs[4])"}
    res = {"workload": f"{names[config]}, h={hidden}, L={L}, {B} windows"}
    st = max(3, min(steps, int(0.06 / {"mck4": 1e-3, "solo": 6e-3, "synth32": 6e-3}[config])))      # steps per block: blocks of ~60 ms
    for plan in ("bf16", PARITY_DTYPE):
        try:
            w = Workload(spec, plan, B, device, 1234)
        except Exception as ex:  # noqa: BLE001
            res[plan if plan == "bf16" else "parity_plan"] = {"error": str(ex)[:200]}
            continue
        med, _ = w.time_blocks(st, max(2, warmup // 2), min_time)
        entry = {"dtype": plan, "ms_per_step": med / st * 1e3, "value": B * st / med}
        stats = w.kernel_stats(st)
        entry["kernel_us"] = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 2) for s in stats}
        r = roofline_of(stats, st, B, plan, w.e, config, L, hidden)
        entry["dominant"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_us", "launches_per_step", "share_of_step", "traffic")}
        if "traffic_source" in r:
            entry["dominant"]["traffic_source"] = r["traffic_source"]
        if "priced_by" in r:
            entry["dominant"]["priced_by"] = r["priced_by"]
        if "operands" in r:      # (HBM-bound kernel: `frac` prices the raw-input bytes only -- SURVEY 8(d); its own operand stream is reported beside it)
            entry["dominant"]["operands_frac"] = r["operands"]["frac"]
        fl = (w.e.info.flops_fwd + w.e.info.flops_bwd) * B
        entry["step_mfma_frac"] = fl / (med / st) / 1e12 / PEAK["mfma_TFLOPs"][plan]
        res["bf16" if plan == "bf16" else "parity_plan"] = entry
        del w
        torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves, as fresh children, BEFORE anything here touches the GPU
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when its communicator is created (seen on the GPU box:
    # "RCCL version : 2.26.6 ..."), so everything written to fd 1 from here on goes to stderr and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    dist = None
    if world > 1 or os.environ.get("MSHGNN_BENCH_FORCE_DIST") == "1":   # (FORCE_DIST: a 1-rank RCCL group, to exercise the N > 1 code path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    B, L, hidden = defaults(args)
    spec = build_spec(L, args.config, hidden)
    wl = Workload(spec, args.dtype, B, device, 1234 + rank, dist, args.overlap, args.grad_exchange, args.collective)
    if args.grad_exchange in ("bf16", "live"):
        wl.can_overlap = False      # (the two-phase step exchanges whole fp32 slices)
    multi = world > 1 or os.environ.get("MSHGNN_BENCH_FORCE_DIST") == "1"
    if args.collective == "auto" and wl.stream_comm is not None and multi:
        wl.calibrate_collective()
    if args.grad_exchange == "auto" and wl.live is not None and multi:
        wl.calibrate_exchange()
    if args.overlap == "auto" and wl.can_overlap and multi:
        live = wl.use_live
        wl.use_live = False
        wl.calibrate_overlap()
        if live and wl.overlap and wl.exchange_choice["ms_live_packed"] <= wl.overlap_choice["ms_two_phase"]:
            wl.overlap = False
            wl.overlap_choice["two_phase"] = False
        wl.use_live = live and not wl.overlap
    med, blocks = wl.time_blocks(args.steps, args.warmup, args.min_time)
    single = None
    if wl.nb > 1 and world == 1:      # the same loop on ONE resident batch (what rounds 1-5 timed), reported beside the headline
        nb_keep, wl.nb = wl.nb, 1
        wl.xs, wl.y = wl.batches[0]
        m1, _ = wl.time_blocks(args.steps, args.warmup, args.min_time / 2)
        single = {"ms_per_step": m1 / args.steps * 1e3, "value": B * args.steps / m1}
        wl.nb = nb_keep
    value = world * B * args.steps / med
    loss = float(wl.loss.item())

    # per-kernel HIP events (separate pass) -> roofline of the dominant kernel
    stats = wl.kernel_stats(args.steps)
    roof = roofline_of(stats, args.steps, B, args.dtype, wl.e, args.config, L, hidden)
    step_s = med / args.steps
    if args.config == "a1c2" and L in (3, 8) and hidden == 128:
        # the WHOLE step against SURVEY.md 8(d): inputs read once forward + once for the encoder's weight gradients
        es = {"bf16": 1.0, "x3": 2.0, "f32": 2.0}[args.dtype]
        sb = 2.0 * SURVEY_8D["bytes_in_bf16"] * es * B
        sf = SURVEY_8D["flops_L3" if L == 3 else "flops_L8"] * B
        roof["step"] = {"algorithmic_bytes": sb, "hbm_frac": sb / step_s / 1e9 / PEAK["hbm_GBs"],
                        "algorithmic_flops": sf, "mfma_frac": sf / step_s / 1e12 / PEAK["mfma_TFLOPs"][args.dtype],
                        "note": "SURVEY 8(d) per-window figures x windows / measured step time (they count every node; see step_live)"}
        # what the plan really has to touch: inputs of the nodes that can reach the output at this depth (twice), FLOPs of the live nodes only
        lb = 2.0 * float(wl.e.info.bytes_in_live) * B
        lf = (wl.e.info.flops_fwd + wl.e.info.flops_bwd) * B
        roof["step_live"] = {"algorithmic_bytes": lb, "hbm_frac": lb / step_s / 1e9 / PEAK["hbm_GBs"],
                             "algorithmic_flops": lf, "mfma_frac": lf / step_s / 1e12 / PEAK["mfma_TFLOPs"][args.dtype],
                             "note": "the same with node-level liveness: only nodes whose values can reach the decoder at this depth (A1-C2 at 3 layers: "
                                     "not the base nodes -- four hops from the feet; the reference computes them and gets exact-zero gradients)"}
    kernels = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 2) for s in stats}

    names = {"a1c2": "A1-C2 GRF regression (3-D)", "mck4": "MiniCheetah-K4 contact classification", "solo": "Solo-12 K4 centroidal-momentum regression (COM_HGNN_K4)",
             "synth32": "synthetic 32-limb MI-HGNN GRF regression", "mcc2": "MiniCheetah-C2 contact classification", "solo_s4": "Solo-12 S4 centroidal-momentum regression (COM_HGNN_S4)", "mi_quad": "MI-HGNN quadruped contact classification (GRF_HGNN)"}
    res = {
        "metric": "graph-windows/sec fwd+bwd, A1-C2 GRF regression" if args.config == "a1c2" else f"graph-windows/sec fwd+bwd, {names[args.config]}",
        "value": value, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{names[args.config]}, h={hidden}, L={L}, T={spec.num_timesteps}, {B} windows/GPU, "
                               f"fwd+{'MSE' if spec.regression else 'CE'}+bwd, all parameter gradients"
                               + (f", RCCL all-reduce (mean over {world} ranks{', overlapped two-phase' if wl.overlap else ''})" if dist is not None else ""),
                   "global_batch": B * world, "parallelism": f"dp{world}", "rccl_ranks": (dist.get_world_size() if dist is not None else 0),
                   **({"grad_exchange": ("live" if wl.use_live else "f32") if args.grad_exchange == "auto" else args.grad_exchange,
                       "grad_exchange_choice": wl.exchange_choice,
                       "collective": "stream (C-ABI ncclAllReduce on the step's stream)" if (wl.use_stream and not wl.overlap) else "torch.distributed",
                       "collective_choice": wl.collective_choice or {"mode": args.collective, "stream": bool(wl.use_stream)}} if dist is not None else {})},
        "timing": {"blocks": len(blocks), "block_steps": args.steps, "median_ms": med * 1e3, "min_ms": min(blocks) * 1e3, "max_ms": max(blocks) * 1e3,
                   "timed_s": sum(blocks)},
        "overlap": wl.overlap_choice or {"mode": args.overlap, "two_phase": bool(wl.overlap)},
        "roofline": roof, "kernel_us": kernels,
        "resident_batches": wl.nb, "single_resident_batch": single,
        "step_kernel": {"specialised": getattr(wl.e, "specialised", ""),
                        "what": "compile-time program the one-call step's stack launch runs on (csrc/mshgnn_spec_tables.inc; '' = the plan tables are interpreted; "
                                "MSHGNN_SPEC=0 forces that -- same bits, tests/test_spec_gpu.py)"},
        "algorithmic_flops_per_window": wl.e.info.flops_fwd + wl.e.info.flops_bwd,
        "flat_gradient_bytes": 4 * spec.flat_size(),
        "loss": loss,
    }
    res["liveness"] = {"node_level": os.environ.get("MSHGNN_PRUNE", "1") != "0",
                       "input_bytes_per_window": float(wl.e.info.bytes_in), "live_input_bytes_per_window": float(wl.e.info.bytes_in_live),
                       "what": "nodes whose output cannot reach the decoder within the model's depth are not computed (exact: the reference gives "
                               "their parameters zero gradients); MSHGNN_PRUNE=0 computes every node of every live type as rounds 1-3 did"}
    extras = rank == 0 and world == 1 and not args.no_extras and args.config == "a1c2" and args.surface == "flat" and hidden == 128
    if rank == 0 and world == 1 and not args.no_extras and args.config == "mck4" and args.dtype == "bf16" and args.surface == "flat" and hidden == 128:
        del wl
        torch.cuda.empty_cache()
        res["end_to_end"] = end_to_end(spec, B, device, args.steps, args.warmup, "mck4")
    if extras:
        # the module surface first, with the headline's workspace still allocated: its two-call route is host-bound on the pool's slower hosts and measured
        # 0.44-0.60 ms/step right after that workspace had been returned to the driver (torch's allocator re-growing its pools), 0.37 in a process that keeps it
        # (`--surface module`) -- the state a training process is in
        res["module_surface"] = module_surface(spec, B, device, args.steps, args.warmup, args.dtype)
        del wl      # free the 0.6 GB workspace before the other side measurements
        torch.cuda.empty_cache()
        # the same workload on the parity-grade plan (north_star tolerance 1e-4), driver-visible
        if args.dtype != PARITY_DTYPE:
            pd = PARITY_DTYPE
            try:
                wp = Workload(spec, pd, B, device, 1234)
            except Exception:  # noqa: BLE001  (plan not available for this topology)
                pd = "f32"
                wp = Workload(spec, pd, B, device, 1234)
            pm, pb = wp.time_blocks(args.steps, args.warmup, args.min_time / 2)
            pk = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 2) for s in wp.kernel_stats(args.steps)}
            del wp
            torch.cuda.empty_cache()
            perr, pflip, pout = parity_error(build_spec(L, args.config, hidden), pd, device)
            res["parity_plan"] = {"dtype": pd, "ms_per_step": pm / args.steps * 1e3, "value": B * args.steps / pm,
                                  "golden": parity_golden(pd, device),
                                  "max_rel_err_vs_oracle": perr, "flipped_relu_decisions": pflip, "relu_decisions_outside_tolerance": pout,
                                  "tolerance": 1e-4, "kernel_us": pk,
                                  "what": "same workload, parity-grade plan.  `golden`: the unconditional check against the committed vectors of the "
                                          "reference run.  max_rel_err_vs_oracle: max-abs error / max-abs reference over every hidden state, output, loss "
                                          "and parameter gradient against the fp64 oracle on 48 seeded windows, the oracle evaluated with the engine's relu "
                                          "decisions (flipped_relu_decisions of them differ from the exact ones, each within 1e-4 of its tensor's scale of zero)"}
        # evaluation: the forward alone, no stashes (training=False) -- what `evaluate_model` / validation_step drive (gnnLightning.py:1004-1100, 724-741)
        res["forward_only"] = {"what": "mshgnn_forward(training=0): encoder + L layers + decoder, no stashes, inputs resident; same windows as the headline"}
        for plan in dict.fromkeys((args.dtype, PARITY_DTYPE)):
            wf = Workload(spec, plan, B, device, 1234)
            dtf = median_step_s(lambda: wf.e.forward(wf.xs, wf.flat, B, training=False, out=wf.out), torch.cuda.synchronize, args.steps, args.warmup)
            res["forward_only"][plan] = {"ms_per_step": dtf * 1e3, "value": B / dtf}
            del wf
            torch.cuda.empty_cache()
        # the other model types the reference's scripts default to, each on its compile-time program (and on the interpreting kernels beside it: MSHGNN_SPEC=0 is read per plan)
        res["other_models"] = {"what": "one-call step (mshgnn_step_mse / _ce), bf16 plan, 8 layers, h=128, 8192 windows, inputs resident: ms per step on the plan's compile-time program / "
                                       "on the interpreting kernels (MSHGNN_SPEC=0)"}
        for cfg_ in ("mcc2", "mi_quad"):
            entry = {}
            for label, env in (("ms_per_step", None), ("interpreted_ms_per_step", "0")):
                prev_ = os.environ.get("MSHGNN_SPEC")
                if env is not None:
                    os.environ["MSHGNN_SPEC"] = env
                try:
                    wo = Workload(build_spec(8, cfg_, 128), "bf16", 8192, device, 1234)
                    mo, _ = wo.time_blocks(10, 3, 0.1)
                    entry[label] = mo / 10 * 1e3
                    if env is None:
                        entry["program"] = wo.e.specialised
                    del wo
                finally:
                    if env is not None:
                        if prev_ is None:
                            del os.environ["MSHGNN_SPEC"]
                        else:
                            os.environ["MSHGNN_SPEC"] = prev_
                torch.cuda.empty_cache()
            res["other_models"][cfg_] = entry
        # the reference's own batch sizes (train_regression-grf_msgn.py:93: 32; train_classification_msgn.py:750 / train_regression-com_msgn.py:76: 64): a step is one
        # tile chain + four short launches -- the one-call step on its compile-time program (the specialised slab kernel at every whole-tile batch size)
        res["small_batches"] = {"what": "mshgnn_step_mse of the headline model at the reference's batch sizes, inputs resident, eager launches (ms per step; windows/s)"}
        for b in (32, 64, 256):
            ws_ = Workload(spec, args.dtype, b, device, 1234)
            dts = median_step_s(ws_.step, torch.cuda.synchronize, max(args.steps, 50), args.warmup, min_time=0.1)
            res["small_batches"][str(b)] = {"ms_per_step": dts * 1e3, "value": b / dts}
            del ws_
        torch.cuda.empty_cache()
        if args.dtype == "bf16" and L == 3:
            res["end_to_end"] = end_to_end(spec, B, device, args.steps, args.warmup)
            torch.cuda.empty_cache()
        if os.environ.get("MSHGNN_PRUNE", "1") != "0":      # the same workload with every node of every live type computed (type-level liveness, rounds 1-3)
            os.environ["MSHGNN_PRUNE"] = "0"
            try:
                wu = Workload(spec, args.dtype, B, device, 1234)
                mu, _ = wu.time_blocks(args.steps, args.warmup, args.min_time / 2)
                ku = {s_["name"]: round(s_["total_ms"] / s_["launches"] * 1e3, 2) for s_ in wu.kernel_stats(args.steps)}
                res["liveness"]["all_nodes_computed"] = {"ms_per_step": mu / args.steps * 1e3, "value": B * args.steps / mu, "kernel_us": ku,
                                                         "algorithmic_flops_per_window": wu.e.info.flops_fwd + wu.e.info.flops_bwd}
                del wu
                torch.cuda.empty_cache()
            finally:
                del os.environ["MSHGNN_PRUNE"]
        if L != 8:      # the paper's depth (train_regression-grf_msgn.py:94)
            w8 = Workload(build_spec(8, args.config, hidden), args.dtype, B, device, 1234)
            m8, _ = w8.time_blocks(args.steps, args.warmup, args.min_time / 2)
            st8 = w8.kernel_stats(args.steps)
            r8 = roofline_of(st8, args.steps, B, args.dtype, w8.e, args.config, 8, hidden)
            s8 = m8 / args.steps
            f8 = (w8.e.info.flops_fwd + w8.e.info.flops_bwd) * B
            b8 = 2.0 * float(w8.e.info.bytes_in_live) * B
            r8["step_live"] = {"algorithmic_bytes": b8, "hbm_frac": b8 / s8 / 1e9 / PEAK["hbm_GBs"], "algorithmic_flops": f8,
                               "mfma_frac": f8 / s8 / 1e12 / PEAK["mfma_TFLOPs"][args.dtype]}
            res["L8"] = {"ms_per_step": s8 * 1e3, "value": B / s8, "dtype": args.dtype,
                         "workload": "A1-C2 GRF regression at the paper's depth (train_regression-grf_msgn.py:94), h=128, L=8, 8192 windows",
                         "kernel_us": {s_["name"]: round(s_["total_ms"] / s_["launches"] * 1e3, 2) for s_ in st8}, "roofline": r8,
                         "algorithmic_flops_per_window": w8.e.info.flops_fwd + w8.e.info.flops_bwd}
            del w8
            torch.cuda.empty_cache()
        if args.dtype == "bf16" and L == 3 and B == 8192:      # the other BASELINE configs, driver-visible
            res["configs"] = {c: side_config(c, device, args.steps, args.warmup) for c in ("mck4", "solo", "synth32")}
    if args.surface == "module" and rank == 0 and world == 1:
        res["module_surface"] = module_surface(spec, B, device, args.steps, args.warmup, args.dtype)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config != "synth32":
        res["cpu_baseline"] = cpu_baseline(spec, args.cpu_batch)
        res["cpu_baseline_B32"] = cpu_baseline(spec, 32, budget_s=6.0, scan=False)     # the reference's own CPU-runnable case
    cfile = os.path.join(ROOT, ".build_commit")      # (tools/profile_round.sh runs: the commit the snapshot was taken at)
    if os.path.exists(cfile):
        res["commit"] = open(cfile).read().strip()
    res["source_hash"] = source_hash()
    if dist is not None:
        try:      # the C-ABI communicator goes first, on every rank, while all of them are still alive
            sc = locals().get("wl") and wl.stream_comm
            if sc:
                sc.close()
        except Exception:  # noqa: BLE001
            pass
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
