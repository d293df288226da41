"""Drop-in model surface of the MS-HGNN hot path: `GRF_HGNN_C2`, `GRF_HGNN_K4`, `GRF_HGNN` with the reference's
constructor signatures, attributes, parameter names and `forward(x_dict, edge_index_dict) -> Tensor` contract
(src/ms_hgnn/lightning_py/hgnn_c2.py:10-12,133; hgnn_k4.py:10-11,146; hgnn.py:10-12,57), so the Lightning wrappers
(gnnLightning.py:564-722) and the research scripts run unchanged on top of it.

All arithmetic runs in the HIP engine behind the C-ABI (engine.py / include/mshgnn.h).  The first forward
(the wrapper's lazy-init call under no_grad, gnnLightning.py:593-595) materialises the lazy encoder weights,
recovers the per-window topology from the PyG-batched `edge_index_dict`, verifies the batch really is B copies
of one graph, and compiles a plan; later calls reuse it.  Precision: `MSHGNN_DTYPE=x3` (default: the split-bf16 parity plan,
<= 1.4e-5 of the fp64 reference against the north_star's 1e-4, at 3.6x the speed of `f32`), `f32` (exact fp32 MFMA, 1e-6) or `bf16`
(throughput mode, 4.6e-3); `model.set_precision("bf16")` switches at run time.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import yaml
from torch import nn

from . import nn as pnn
from .spec import ModelSpec, rel_key
from .engine import accepted_pitches
from .topology import NODE_TYPES, RobotTopology, infer_window_edges


class _Flatten(torch.autograd.Function):
    """named parameters -> the C-ABI's flat fp32 buffer; backward hands each parameter a view of the flat
    gradient (cast back to the parameter's dtype)."""

    @staticmethod
    def forward(ctx, size, device, offsets, *params):
        ctx.offsets = offsets
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        flat = torch.zeros(size, dtype=torch.float32, device=device)
        views = [flat[o:o + n].view(p.shape) for (o, n), p in zip(offsets, params)]
        torch._foreach_copy_(views, [p.detach() for p in params])
        return flat

    @staticmethod
    def backward(ctx, gflat):
        grads = []
        for (o, n), (shape, dt, dev) in zip(ctx.offsets, ctx.meta):
            grads.append(gflat[o:o + n].view(shape).to(device=dev, dtype=dt))
        return (None, None, None, *grads)


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, engine, B, training, *xs):
        out = engine.forward(xs, flat, B, training=training)
        ctx.engine, ctx.B, ctx.xs, ctx.flat = engine, B, xs, flat
        ctx.ticket = engine.stash_ticket(B)
        return out

    @staticmethod
    def backward(ctx, gout):
        e = ctx.engine
        if e.stash_ticket(ctx.B) != ctx.ticket:
            raise RuntimeError("the activation stash of this forward was overwritten by a later forward of the same "
                               "batch size on the same engine; call backward before the next forward")
        gflat = e.backward(ctx.xs, ctx.flat, gout.contiguous().to(torch.float32), ctx.B)
        return (gflat, None, None, None, *([None] * len(ctx.xs)))


class _EngineFnP(torch.autograd.Function):
    """The hot path for parameters that already live as views of the engine's flat fp32 buffer (`_MSHGNNBase._flat_params`): nothing
    is copied on the way in, and backward hands every parameter a VIEW of the fresh flat gradient buffer the C-ABI fills -- no
    per-parameter cast or copy kernels (the parameters are autograd inputs only so that their .grad gets populated)."""

    @staticmethod
    def forward(ctx, engine, B, flat, offsets, ddp, n_x, *args):
        """ddp: None, or the process group (True: the default one) of ddp.flat_data_parallel -- the flat gradient is then all-reduced here, since
        no DistributedDataParallel wrapper hooks the parameters on that route."""
        xs = args[:n_x]
        out = engine.forward(xs, flat, B, training=True)
        ctx.engine, ctx.B, ctx.xs, ctx.flat, ctx.offsets, ctx.n_x, ctx.ddp = engine, B, xs, flat, offsets, n_x, ddp
        ctx.shapes = [p.shape for p in args[n_x:]]
        ctx.ticket = engine.stash_ticket(B)
        return out

    @staticmethod
    def backward(ctx, gout):
        e = ctx.engine
        if e.stash_ticket(ctx.B) != ctx.ticket:
            raise RuntimeError("the activation stash of this forward was overwritten by a later forward of the same "
                               "batch size on the same engine; call backward before the next forward")
        n_flat = ctx.flat.numel()
        full = torch.empty(n_flat + 16, dtype=torch.float32, device=ctx.flat.device)      # a fresh buffer per backward: .grad views never alias
        gflat = e.backward(ctx.xs, ctx.flat, gout.contiguous().to(torch.float32), ctx.B, grad_flat=full[:n_flat])
        if ctx.ddp is not None:      # ddp.flat_data_parallel on the two-call route: the same single exchange the fused training step makes
            from .ddp import exchange_flat_gradient_
            group, weighted, live = (*ctx.ddp, None)[:3]
            gflat.div_(exchange_flat_gradient_(full, n_flat, ctx.B, None if group is True else group, weighted, live))
        grads = [gflat[o:o + n].view(shape) for (o, n), shape in zip(ctx.offsets, ctx.shapes)]
        return (None, None, None, None, None, None, *([None] * ctx.n_x), *grads)


class _EngineFnFast(torch.autograd.Function):
    """Single-process fast path: like _EngineFnP, but the parameter gradients do not travel through autograd at all -- 50 AccumulateGrad
    nodes and 50 freshly sliced views per step cost more host time than the whole step takes on the GPU.  The C-ABI writes the flat
    gradient into a persistent buffer whose per-parameter views are cached; backward assigns them to .grad when the gradients were
    cleared (zero_grad(set_to_none=True), the first step) and ADDS a fresh gradient in place otherwise (accumulation, or gradients
    zeroed in place), which is what autograd's accumulation would have produced.  Not used under torch.distributed (DDP needs the
    per-parameter autograd hooks: _EngineFnP)."""

    @staticmethod
    def forward(ctx, anchor, model, engine, B, *xs):
        out = engine.forward(xs, model._flat, B, training=True)
        ctx.model, ctx.engine, ctx.B, ctx.xs, ctx.flat = model, engine, B, xs, model._flat
        ctx.ticket = engine.stash_ticket(B)
        return out

    @staticmethod
    def backward(ctx, gout):
        e, m = ctx.engine, ctx.model
        if e.stash_ticket(ctx.B) != ctx.ticket:
            raise RuntimeError("the activation stash of this forward was overwritten by a later forward of the same "
                               "batch size on the same engine; call backward before the next forward")
        g32 = gout.contiguous().to(torch.float32)
        _deliver_gradients(m, ctx.flat, lambda target: e.backward(ctx.xs, ctx.flat, g32, ctx.B, grad_flat=target))
        return (None, None, None, None, *([None] * len(ctx.xs)))


def _deliver_gradients(m, flat, fresh, scale=None):
    """Hand a freshly computed flat gradient to the parameters' .grad (views of the persistent m._gflat): assigned when the gradients were
    cleared (zero_grad(set_to_none=True), the first step), ADDED otherwise -- what autograd's accumulation would have produced.  Decided per
    TRAINABLE parameter: a frozen parameter's .grad stays None and says nothing; a .grad that is not this buffer's view (left over from
    before a .to() re-created the flat buffers, or assigned by another route) is accumulated into in place, like autograd would.
    fresh: callable(target | None) -> the flat gradient, written into `target` when given."""
    params = m._param_list
    if m._gflat is None or m._gflat.device != flat.device:
        m._gflat = torch.zeros_like(flat)
        m._gviews = [m._gflat[o:o + n].view(p.shape) for (o, n), p in zip(m._spec.param_offsets().values(), params)]
    trainable = [(p, v) for p, v in zip(params, m._gviews) if p.requires_grad]
    if all(p.grad is None for p, _ in trainable):      # every gradient was cleared: the buffer is free to be overwritten
        fresh(m._gflat)
        if scale is not None:
            m._gflat.mul_(scale)
        for p, v in trainable:
            p.grad = v
        return
    g = fresh(None)
    if scale is not None:
        g = g * scale
    if all(p.grad is v for p, v in trainable):         # accumulation into the views of this buffer: one flat add
        m._gflat.add_(g)
        return
    for (p, v), (o, n) in zip(zip(params, m._gviews), m._spec.param_offsets().values()):
        if not p.requires_grad:
            continue
        gp = g[o:o + n].view(p.shape)
        if p.grad is None:           # cleared individually: takes the view, holding exactly this step's gradient
            v.copy_(gp)
            p.grad = v
        elif p.grad is v:
            v.add_(gp)
        else:                        # a foreign gradient tensor (stale view of an earlier buffer, a user-assigned tensor): accumulate in place
            p.grad.add_(gp.to(p.grad.dtype))


class _FusedStepFn(torch.autograd.Function):
    """_MSHGNNBase.fused_training_step: the whole step (forward, wrapper loss, backward) runs in forward() as one engine call that leaves the
    flat gradient in a pending buffer; backward() hands it to the parameters times the upstream factor (Lightning clears the gradients
    BETWEEN training_step and backward, so nothing may touch .grad before).  target: fp32 labels (regression) or int32 contact flags."""

    @staticmethod
    def forward(ctx, anchor, model, run):
        """run(grad_flat) -> (out, loss[1]): the engine call (mshgnn_step_mse / _ce or their *_series forms)."""
        if model._gpend is None or model._gpend.device != model._flat.device:
            n = model._flat.numel()      # (+ spare elements: the window count of ddp.exchange_flat_gradient_ rides behind the gradient)
            model._gpend_full = torch.empty(n + 16, dtype=torch.float32, device=model._flat.device)
            model._gpend = model._gpend_full[:n]
        out, loss = run(model._gpend)
        model._gpend_id += 1
        ctx.model, ctx.flat, ctx.ticket = model, model._flat, model._gpend_id
        ctx.windows = out.shape[0] // max(1, model._spec.num_nodes[model._spec.out_type])
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # (no zero tensor for the output's absent gradient)
        return loss[0], out

    @staticmethod
    def backward(ctx, gl, _gout):
        m = ctx.model
        n_in = len(ctx.needs_input_grad)
        if gl is None:
            return (None,) * n_in
        if m._gpend_id != ctx.ticket:
            raise RuntimeError("the pending gradient of this training step was overwritten by a later fused training step of the same "
                               "model; call backward before the next step")
        pend, scale = m._gpend, gl.to(device=m._gpend.device, dtype=torch.float32)
        if m._flat_ddp is not None:      # data parallel: ONE sum-all-reduce of the flat gradient (weighted by the ranks' window counts: ddp.py)
            from .ddp import exchange_flat_gradient_
            group = None if m._flat_ddp is True else m._flat_ddp
            scale = scale / exchange_flat_gradient_(m._gpend_full, pend.numel(), ctx.windows, group, m._flat_ddp_weighted, getattr(m, "_flat_ddp_live", None))
        _deliver_gradients(m, ctx.flat, lambda target: torch.mul(pend, scale, out=target) if target is not None else pend * scale)
        return (None,) * n_in


class _MSHGNNBase(nn.Module):
    kind = "c2"
    num_bases = 2

    def _init_common(self, hidden_channels, num_layers, data_metadata, regression, activation_fn):
        # The fused engines implement the reference's default, nn.ReLU().  Any other activation module
        # (the constructors accept one, hgnn_c2.py:10-12) runs the same forward operator by operator on the stand-alone HIP operators
        # of ops.py (_forward_operators): PyG-style launches instead of the fused kernels -- slower, same numerics, still no CPU path.
        self._fused_activation = isinstance(activation_fn, nn.ReLU)      # (any hidden width: the engines' widths are multiples of 128, other widths run
                                                                       #  zero-padded to the next one -- engine.PaddedEngine -- with identical results)
        self.regression = regression
        self.activation = activation_fn
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self._node_types = list(data_metadata[0])
        self._edge_types = [tuple(e) for e in data_metadata[1]]
        self._engines: Dict[Tuple[str, str], object] = {}
        self._spec: Optional[ModelSpec] = None
        self._flat: Optional[torch.Tensor] = None
        self._flat_ok = False            # parameters are fp32 views into self._flat (device-resident fast path)
        self._param_list = None
        self._gflat = None               # persistent flat gradient + its cached per-parameter views (_EngineFnFast)
        self._gpend = None               # the flat gradient a fused training step computed, until its backward() delivers it
        self._gpend_id = 0
        self._flat_ddp = None            # ddp.flat_data_parallel: the process group the fused training step all-reduces its flat gradient over
        self._flat_ddp_weighted = False  # ... True: each rank's gradient weighted by its window count (opt-in; the default is DDP's mean of the ranks' means)
        self._flat_ddp_live = None       # ... over the live elements only (flat_data_parallel(live_only=True)): (index, packed buffer)
        self._gpend_full = None
        self._gviews = None
        self._anchor = None
        self._checked_batches = set()
        self._precision = os.environ.get("MSHGNN_DTYPE", "x3")
        self._group = None

    # ---- copy / pickle: compiled plans (ctypes handles) and device scratch never travel; they are rebuilt lazily ----
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"] = {}
        state["_flat"] = None
        state["_flat_ok"] = False
        state["_gflat"] = state["_gviews"] = state["_anchor"] = state["_gpend"] = state["_gpend_full"] = state["_flat_ddp"] = state["_flat_ddp_live"] = None
        state["_param_list"] = None
        state["_checked_batches"] = set()
        return state

    def zero_grad(self, set_to_none: bool = True) -> None:
        """nn.Module.zero_grad walks the whole module tree (~0.3 ms of host time for these ~80 sub-modules: as long as the GPU step);
        same semantics over the cached parameter list."""
        if self._spec is None:
            return super().zero_grad(set_to_none)
        for p in self._params_in_flat_order():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:      # (torch.optim's zero_grad: .grad tensors here are views of the flat gradient buffer, which detach_() refuses)
                    if p.grad.grad_fn is not None:
                        p.grad.detach_()
                    else:
                        p.grad.requires_grad_(False)
                    p.grad.zero_()

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .double() / .float() replace the parameter tensors: the flat-buffer views are re-established by the next forward
        self._flat_ok = False
        return super()._apply(fn, *args, **kwargs)

    def _flat_params(self, pdev):
        """Device-resident fast path: every parameter's storage IS a slice of one flat fp32 buffer in state_dict order (the layout the
        C-ABI takes), so a forward copies nothing.  Parameters therefore live in fp32 on the device whatever the default dtype is (the
        engine's master weights are fp32; optimizers update the views in place; load_state_dict copies into them).  A later
        .to() / .double() un-does this; the next forward re-establishes it."""
        if self._flat_ok and self._flat is not None and self._flat.device == pdev:
            return self._flat
        params = self._params_in_flat_order()
        flat = torch.zeros(self._spec.flat_size(), dtype=torch.float32, device=pdev)
        with torch.no_grad():
            for (o, n), p in zip(self._spec.param_offsets().values(), params):
                v = flat[o:o + n].view(p.shape)
                v.copy_(p.detach())
                p.data = v
        self._flat, self._flat_ok = flat, True
        self._gflat = self._gviews = self._gpend = None
        self._anchor = torch.zeros((), device=pdev, requires_grad=True)
        return flat

    def _params_in_flat_order(self):
        if self._param_list is None:
            sd = dict(self.named_parameters())
            self._param_list = [sd[k] for k in self._spec.param_offsets().keys()]
        return self._param_list

    # ---- construction helpers ---------------------------------------------------------------------
    def _build_convs(self, mean_rels):
        self.encoder = pnn.HeteroDictLinear(-1, self.hidden_channels, self._node_types)
        self.convs = nn.ModuleList()
        h = self.hidden_channels
        for _ in range(self.num_layers):
            conv_dict = {}
            for et in self._edge_types:
                conv_dict[et] = pnn.GraphConv(h, h, aggr="mean" if et[1] in mean_rels else "add")
            self.convs.append(pnn.HeteroConv(conv_dict, aggr="sum"))

    def set_precision(self, dtype: str):
        if dtype not in ("f32", "bf16", "x3"):
            raise ValueError("precision must be 'f32', 'bf16' or 'x3'")
        self._precision = dtype
        return self

    def reset_parameters(self):
        """Reset all learnable parameters (hgnn_c2.py:286-293)."""
        self.encoder.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        if hasattr(self, "base_transform"):
            self.base_transform[0].reset_parameters()
            self.base_transform[2].reset_parameters()
        self.decoder.reset_parameters()

    # ---- plan ---------------------------------------------------------------------------------------
    def _num_nodes(self, x_dict) -> Tuple[int, Dict[str, int]]:
        nb = self.num_bases
        if x_dict["base"].shape[0] % nb:
            raise ValueError(f"x_dict['base'] has {x_dict['base'].shape[0]} rows, not a multiple of {nb} base nodes")
        B = x_dict["base"].shape[0] // nb
        nn_ = {}
        for t in self._node_types:
            if x_dict[t].shape[0] % B:
                raise ValueError(f"x_dict['{t}'] rows ({x_dict[t].shape[0]}) are not a multiple of the batch size {B}")
            nn_[t] = x_dict[t].shape[0] // B
        return B, nn_

    def _make_spec(self, x_dict, edge_index_dict) -> ModelSpec:
        B, nn_ = self._num_nodes(x_dict)
        rels = []
        for et in self._edge_types:
            if et not in edge_index_dict:
                raise ValueError(f"edge_index_dict lacks relation {et} named in data_metadata")
            s, _, d = et
            rels.append((et, infer_window_edges(edge_index_dict[et], nn_[s], nn_[d], B)))
        topo = RobotTopology(name="from-batch", num_nodes={t: nn_[t] for t in NODE_TYPES if t in nn_}, relations=rels)
        widths = {t: int(x_dict[t].shape[1]) for t in self._node_types}
        return ModelSpec(kind=self.kind, topology=topo, hidden=self.hidden_channels, num_layers=self.num_layers,
                         widths=widths, regression=self.regression, grf_dimension=getattr(self, "grf_dimension", 1),
                         group=self._group, num_timesteps=getattr(self, "num_timesteps", 150),
                         com_dimension=getattr(self, "num_dimensions_per_base", 6) if self.kind == "s4_com" else 6)

    def _engine(self, device):
        from .engine import make_engine
        key = (self._precision, str(device))
        if key not in self._engines:
            self._engines[key] = make_engine(self._spec, dtype=self._precision, device=device)
        return self._engines[key]

    # ---- forward --------------------------------------------------------------------------------------
    def _prepare(self, x_dict, edge_index_dict):
        """Checks, lazy initialisation and the engine for this call: (spec, B, engine | None, device of the parameters' engine copy)."""
        for t in self._node_types:
            if t not in x_dict:
                raise KeyError(f"x_dict lacks node type '{t}'")
        first = self._spec is None
        if first:
            for t in self._node_types:
                self.encoder.lins[t].materialize(int(x_dict[t].shape[1]))
                lin = self.encoder.lins[t]
                ref = self.decoder.weight
                lin.to(device=ref.device, dtype=ref.dtype)
            self._spec = self._make_spec(x_dict, edge_index_dict)
            expect = list(self._spec.param_shapes().keys())
            have = [k for k, _ in self.named_parameters()]
            if expect != have:
                raise RuntimeError(f"parameter layout mismatch: {set(expect) ^ set(have)}")
        spec = self._spec
        B, nn_ = self._num_nodes(x_dict)
        def _pitch_ok(x, F):     # the reference's width, or rows already at an engine pitch (on-device window assembly): whole 16-byte chunks >= F
            return x.shape[1] == F or (x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] in accepted_pitches(F, x.element_size()))
        for t in self._node_types:
            if nn_[t] != spec.num_nodes[t] or not _pitch_ok(x_dict[t], spec.widths[t]):
                raise ValueError(f"x_dict['{t}'] does not match the compiled topology "
                                 f"({nn_[t]} nodes x {x_dict[t].shape[1]} vs {spec.num_nodes[t]} x {spec.widths[t]})")
        if B not in self._checked_batches:   # one host-side check per batch size: B copies of the compiled graph
            for et in self._edge_types:
                s, _, d = et
                if infer_window_edges(edge_index_dict[et], nn_[s], nn_[d], B) != spec.topology.edges(et):
                    raise ValueError(f"edge_index_dict[{et}] differs from the topology this model was compiled for")
            self._checked_batches.add(B)
        if not self._fused_activation:
            return spec, B, None, None
        pdev = self.decoder.weight.device
        if pdev.type != "cuda":
            # parameters still on the host (e.g. the wrapper's lazy-init call, gnnLightning.py:593-595, or
            # evaluate_model's model.to('cpu'), :1029): the arithmetic STILL runs in the HIP engine, on the current
            # device, with a device copy of the parameters.  No GPU -> error; there is no CPU implementation.
            if not torch.cuda.is_available():
                raise RuntimeError("the MS-HGNN engine needs a HIP device; there is no CPU fallback")
            pdev = torch.device("cuda", torch.cuda.current_device())
        return spec, B, self._engine(pdev), pdev

    def _shape_output(self, out, spec, B, in_dev, in_dtype):
        out = out.to(device=in_dev, dtype=in_dtype if in_dtype in (torch.float64, torch.float32) else torch.float32)
        if spec.output_is_window_major:
            return out.view(B, -1)      # ms_foot_decoder: [B, 4*3]   (hgnn_c2.py:184-189)
        if spec.kind in ("k4_com", "c2_com"):
            return out.view(B, spec.num_nodes["base"], spec.out_channels)   # morphological_symmetry_decoder, hgnn_k4_com.py:157-165
        return out                      # [B*4, out_channels_per_foot]  (COM S4 / COM_HGNN: [B, com_dimension])

    def fused_training_step(self, x_dict, edge_index_dict, y):
        """forward + the wrapper's loss + backward in ONE engine call (mshgnn_step_mse / mshgnn_step_ce: the decoder, the loss and the decoder's
        backward run in the tail of the fused forward kernel) for a caller that has the labels at forward time -- the training-step wrappers.
        y: the batch's labels (regression: B * n_out * out_channels values; classification: B * 4 contact flags).  Returns (out, loss):
        `out` as forward() returns it (no autograd), `loss` a scalar whose backward() delivers the parameter gradients the engine has
        already computed (times the upstream factor) -- same values as forward() + loss + backward() through autograd.
        Returns None when this route does not apply (gradients disabled, parameters on the host, the operator-by-operator path, more than
        one process without `ddp.flat_data_parallel`: torch's DDP needs the per-parameter autograd hooks); the caller then takes the two-call route."""
        spec, B, e, pdev = self._prepare(x_dict, edge_index_dict)
        params = self._params_in_flat_order()
        if e is None or not torch.is_grad_enabled() or params[0].device.type != "cuda" or not all(p.requires_grad for p in params):
            return None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self._flat_ddp is None:
            return None
        in_dev, in_dtype = x_dict[self._node_types[0]].device, x_dict[self._node_types[0]].dtype
        xs = e.cast_inputs(x_dict)
        self._flat_params(pdev)
        if spec.regression:
            target = y.detach().to(pdev, torch.float32).flatten().contiguous()
        else:
            target = (y.detach().to(pdev) != 0).to(torch.int32).flatten().contiguous()
        step = e.step_mse if spec.regression else e.step_ce
        loss, out = _FusedStepFn.apply(self._anchor, self, lambda g: step(xs, self._flat, target, B, grad_flat=g)[:2])
        return self._shape_output(out, spec, B, in_dev, in_dtype), loss

    def fused_training_step_windows(self, batch):
        """`fused_training_step` for a `windows.WindowBatch`: the encoder gathers the batch's inputs straight from the sequence's resident
        series (mshgnn_step_mse_series / mshgnn_step_ce_series -- no separate assembly pass over the windows); the labels (and the
        materialised windows) are left on the batch.  Returns (out, loss) or None when this route does not apply: the model has not seen
        its lazy-initialising forward yet, a plan other than bf16 / split-bf16 on the LDS-resident kernels, a store whose dtype is not the plan's
        input dtype or that is not fast-layout / unstandardised, or whose recipe differs from the model's node types and widths, plus fused_training_step's own conditions."""
        spec = self._spec
        store, B = batch.store, batch.batch_size
        r = store.recipe
        if spec is None or not self._fused_activation or store.dtype not in ("bf16", "x3", "f32") or r.normalize or not store.desc.fast_layout or not r.label_cols:
            return None
        if list(r.node_types) != list(spec.node_types) or any(r.num_nodes[t] != spec.num_nodes[t] or r.width(t) != spec.widths[t] for t in r.node_types):
            return None
        params = self._params_in_flat_order()
        if not torch.is_grad_enabled() or params[0].device.type != "cuda" or params[0].device != store.device or not all(p.requires_grad for p in params):
            return None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self._flat_ddp is None:
            return None
        e = self._engine(store.device)
        if e.generic or getattr(e, "padded", False) or e.storage not in ("bf16", "x3") or (e.storage == "bf16") != (store.dtype == "bf16"):      # (the split plan gathers fp32 series)
            return None
        if B not in self._checked_batches:   # one host-side check per batch size: B copies of the compiled graph
            for et in self._edge_types:
                s_, _, d_ = et
                if infer_window_edges(batch.edge_index_dict[et], spec.num_nodes[s_], spec.num_nodes[d_], B) != spec.topology.edges(et):
                    raise ValueError(f"edge_index_dict[{et}] differs from the topology this model was compiled for")
            self._checked_batches.add(B)
        self._flat_params(store.device)
        step = e.step_mse_series if spec.regression else e.step_ce_series

        def run(g):
            xs, y, out, loss, _ = step(store, batch.starts, self._flat, grad_flat=g)
            _, yf, q = store._buffers(B)
            batch._labels_from_step(xs, yf, q)
            return out, loss
        loss, out = _FusedStepFn.apply(self._anchor, self, run)
        return self._shape_output(out, spec, B, store.device, torch.float32), loss

    def forward(self, x_dict, edge_index_dict):
        spec, B, e, pdev = self._prepare(x_dict, edge_index_dict)
        if e is None:
            return self._forward_operators(x_dict, edge_index_dict, B)
        in_dev, in_dtype = x_dict[self._node_types[0]].device, x_dict[self._node_types[0]].dtype
        xs = e.cast_inputs(x_dict)
        params = self._params_in_flat_order()
        offsets = list(spec.param_offsets().values())
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if params[0].device.type == "cuda":
            # parameters on the device: they are (made) views of the flat fp32 buffer -- no copy in, no copy out
            flat = self._flat_params(pdev)
            if need_grad:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    out = _EngineFnP.apply(e, B, flat, offsets, None if self._flat_ddp is None else (self._flat_ddp, self._flat_ddp_weighted, getattr(self, "_flat_ddp_live", None)), len(xs), *xs, *params)      # gradients through autograd: DDP's hooks see them
                else:
                    out = _EngineFnFast.apply(self._anchor, self, e, B, *xs)
            else:
                with torch.no_grad():
                    out = e.forward(xs, flat, B, training=False)
        else:
            # parameters on the host: a device copy per forward (the slow path: lazy-init call, CPU-pinned evaluation)
            if self._flat is None or self._flat.device != pdev or self._flat_ok:
                self._flat, self._flat_ok = torch.zeros(spec.flat_size(), dtype=torch.float32, device=pdev), False
            if need_grad:
                flat = _Flatten.apply(spec.flat_size(), pdev, offsets, *params)
                out = _EngineFn.apply(flat, e, B, True, *xs)
            else:
                with torch.no_grad():
                    views = [self._flat[o:o + n].view(p.shape) for (o, n), p in zip(offsets, params)]
                    torch._foreach_copy_(views, [p.detach() for p in params])
                    out = e.forward(xs, self._flat, B, training=False)
        return self._shape_output(out, spec, B, in_dev, in_dtype)

    def _forward_operators(self, x_dict, edge_index_dict, B):
        """The reference's forward (hgnn_c2.py:133-182, hgnn_k4.py:146-196, hgnn.py:57-62, hgnn_*_com.py) operator by operator, for models built with
        an activation_fn other than nn.ReLU(): every Linear / GraphConv runs on the HIP operators of ops.py with autograd, the activation itself
        is the caller's module.  Inputs and parameters must be on the device."""
        from . import ops
        spec, act = self._spec, self.activation
        dev = x_dict[self._node_types[0]].device
        if dev.type != "cuda":
            raise RuntimeError("the operator path of a non-ReLU model runs on a HIP device (inputs are on the host); there is no CPU fallback")
        if self.decoder.weight.device != dev:
            raise RuntimeError("move the model to the inputs' device first (model.to(device))")
        x = {}
        for t, m in spec.input_masks().items():                # apply_symmetry (hgnn_c2.py:191-284) as its net +-1 masks
            v = x_dict[t]
            if v.shape[1] != spec.widths[t]:
                raise ValueError(f"x_dict['{t}'] must have the reference's width {spec.widths[t]} on the operator path")
            n = m.shape[0]
            x[t] = (v.view(-1, n, v.shape[1]) * m.to(v.device, v.dtype).unsqueeze(0)).reshape(v.shape)
        x = {k: act(v) for k, v in self.encoder(x).items()}
        bt = getattr(self, "base_transform", None)
        for conv in self.convs:
            h = conv(x, edge_index_dict)
            if spec.kind in ("mi", "s4_com") or bt is None:
                x = {k: act(v) for k, v in h.items()}
                continue
            new = {}
            for k, v in h.items():
                if k == "base":                                   # base_transform: Linear, nn.ReLU() (fixed in the reference), Linear
                    new[k] = ops.linear(torch.relu(ops.linear(v, bt[0].weight, bt[0].bias)), bt[2].weight, bt[2].bias)
                else:
                    new[k] = act(v)
            x = {k: new[k] + x[k] if (k in x and x[k].shape == new[k].shape) else new[k] for k in new}
        out = self.decoder(x[spec.out_type])
        n_out = spec.num_nodes[spec.out_type]
        out = (out.view(B, n_out, -1) * spec.output_mask().to(out.device, out.dtype).unsqueeze(0)).reshape(B * n_out, -1)
        if spec.output_is_window_major:
            return out.view(B, -1)
        if spec.kind in ("k4_com", "c2_com"):
            return out.view(B, spec.num_nodes["base"], spec.out_channels)
        return out

    # ---- reference helper kept for API compatibility ------------------------------------------------------
    def apply_symmetry(self, x_dict):
        """Elementwise +-1 masks equivalent to the reference's apply_symmetry (hgnn_c2.py:191-231).  The engine
        folds these into its encoder loads; this method exists for callers that use it directly."""
        spec = self._spec if self._spec is not None else self._make_spec(x_dict, {})
        masks = spec.input_masks()
        for t, m in masks.items():
            x = x_dict[t]
            n = m.shape[0]
            x_dict[t] = (x.view(-1, n, x.shape[1]) * m.to(x.device, x.dtype).unsqueeze(0)).reshape(x.shape)
        return x_dict


class _SymmetryHelpers:
    """The tensor helpers the reference's C2 / K4 models carry next to `apply_symmetry` (hgnn_c2.py:184-189, 233-284; hgnn_k4.py): the engine needs
    none of them (the masks are sign-bit XORs inside its encoder, the output mask sits in its decoder tail), they exist for callers that use them."""

    def ms_foot_decoder(self, x):
        """[B * num_legs, out_channels_per_foot] decoder rows -> [B, num_legs * out_channels_per_foot] times the feet's +-1 mask (hgnn_c2.py:184-189)."""
        x = x.view(-1, self.num_legs, self.out_channels_per_foot).flatten(start_dim=1)
        return self.feet_linear_weights.to(x.device) * x

    def unpack_data(self, data, batch_size, num_nodes):
        """Node rows [batch * nodes, 2 * dims * T] laid out [variable][axis][time] -> the two variables as [batch, T, nodes * dims]
        (hgnn_c2.py:233-264)."""
        d, t = self.num_dimensions_per_foot, self.num_timesteps
        x = data.view(batch_size, num_nodes, 2, d, t).permute(2, 0, 4, 1, 3)           # [variable, batch, time, node, axis]
        return x[0].reshape(batch_size, t, num_nodes * d), x[1].reshape(batch_size, t, num_nodes * d)

    def pack_data(self, f_p, f_v, batch_size, num_nodes):
        """The inverse of `unpack_data` (hgnn_c2.py:266-284)."""
        d, t = self.num_dimensions_per_foot, self.num_timesteps
        both = torch.stack([f_p.view(batch_size, t, num_nodes, d), f_v.view(batch_size, t, num_nodes, d)])      # [variable, batch, time, node, axis]
        return both.permute(1, 3, 0, 4, 2).reshape(batch_size * num_nodes, 2 * d * t)


def _load_group(symmetry_mode, group_operator_path):
    if symmetry_mode and group_operator_path:
        with open(group_operator_path, "r") as f:
            return yaml.safe_load(f)
    return None


class GRF_HGNN_C2(_SymmetryHelpers, _MSHGNNBase):
    """MS-HGNN for the C2 graph (2 base nodes) -- drop-in for hgnn_c2.py:GRF_HGNN_C2."""
    kind = "c2"
    num_bases = 2

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), symmetry_mode: str = None, group_operator_path: str = None,
                 grf_dimension: int = 3):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.num_timesteps = 150           # hard-coded in the reference (hgnn_c2.py:30)
        self.num_legs = 4
        self.num_joints = 12
        self.num_dimensions_per_foot = 3
        self.num_dimensions_per_base = 3
        self.num_variables_per_joint = 3 if regression else 2
        self.grf_dimension = grf_dimension
        self._group = _load_group(symmetry_mode, group_operator_path)
        self._build_convs(mean_rels=("center_bb",))
        h = hidden_channels
        self.base_transform = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, h))
        if regression and grf_dimension == 1:
            self.out_channels_per_foot = 1
        elif regression and grf_dimension == 3:
            self.out_channels_per_foot = 3
        else:
            self.out_channels_per_foot = 2
        self.decoder = pnn.Linear(h, self.out_channels_per_foot)
        coeffs = ModelSpec.symmetry_coefficients(self._coeff_probe())
        self.joints_linear_weights, self.feet_linear_weights, self.base_coefficients_lin, self.base_coefficients_ang = coeffs

    def _coeff_probe(self):
        class _P:  # minimal duck-type for ModelSpec.symmetry_coefficients
            pass
        p = _P()
        p.kind, p.group, p.num_nodes = self.kind, self._group, {"base": self.num_bases}
        return p


class GRF_HGNN_K4(_SymmetryHelpers, _MSHGNNBase):
    """MS-HGNN for the K4 graph (4 base nodes) -- drop-in for hgnn_k4.py:GRF_HGNN_K4."""
    kind = "k4"
    num_bases = 4

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), symmetry_mode: str = None, group_operator_path: str = None):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.num_timesteps = 150           # hgnn_k4.py:29
        self.num_legs = 4
        self.num_joints = 12
        self.num_dimensions_per_foot = 3
        self.num_dimensions_per_base = 3
        self._group = _load_group(symmetry_mode, group_operator_path)
        self._build_convs(mean_rels=("gt", "gs"))
        h = hidden_channels
        self.base_transform = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, h))
        self.out_channels_per_foot = 1 if regression else 2
        self.decoder = pnn.Linear(h, self.out_channels_per_foot)
        coeffs = ModelSpec.symmetry_coefficients(GRF_HGNN_C2._coeff_probe(self))
        self.joints_linear_weights, self.feet_linear_weights, self.base_coefficients_lin, self.base_coefficients_ang = coeffs


class GRF_HGNN(_MSHGNNBase):
    """MI-HGNN baseline (1 base node, no masks / base MLP / residual) -- drop-in for hgnn.py:GRF_HGNN."""
    kind = "mi"
    num_bases = 1

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), grf_dimension: int = 1):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.grf_dimension = grf_dimension
        self._build_convs(mean_rels=())
        if regression and grf_dimension == 1:
            self.out_channels_per_foot = 1
        elif regression and grf_dimension == 3:
            self.out_channels_per_foot = 3
        else:
            self.out_channels_per_foot = 2
        self.decoder = pnn.Linear(hidden_channels, self.out_channels_per_foot)


class _COMSymBase(_MSHGNNBase):
    """Shared constructor of the Solo centroidal-momentum MS-HGNNs: joints carry (q, qd) of one time step, the decoder
    reads the base nodes and emits [lin(3) | ang(3)] per base node (hgnn_k4_com.py:22-123, hgnn_c2_com.py:22-108)."""
    _mean_rels = ("gt", "gs")

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), symmetry_mode: str = None, group_operator_path: str = None):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.num_timesteps = 1             # hgnn_k4_com.py:29
        self.num_legs = 4
        self.num_joints = 12
        self.num_dimensions_per_base = 6
        self._group = _load_group(symmetry_mode, group_operator_path)
        self._build_convs(mean_rels=self._mean_rels)
        h = hidden_channels
        self.base_transform = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, h))
        self.decoder = pnn.Linear(h, self.num_dimensions_per_base)
        coeffs = ModelSpec.symmetry_coefficients(GRF_HGNN_C2._coeff_probe(self))
        self.joints_linear_weights, _, self.base_coefficients_lin, self.base_coefficients_ang = coeffs

    def morphological_symmetry_decoder(self, x):
        """The reference's output mask as a standalone helper (hgnn_k4_com.py:157-165); forward() already applies it
        inside the decoder kernel."""
        nb = self.num_bases
        x = x.reshape(-1, nb, 6)
        m = torch.cat((self.base_coefficients_lin.view(nb, 3), self.base_coefficients_ang.view(nb, 3)), dim=1)
        return x * m.to(x.device, x.dtype).unsqueeze(0)


class COM_HGNN_K4(_COMSymBase):
    """Centroidal-momentum MS-HGNN on the Solo K4 graph (4 base nodes) -- drop-in for hgnn_k4_com.py:COM_HGNN_K4."""
    kind = "k4_com"
    num_bases = 4


class COM_HGNN_C2(_COMSymBase):
    """Centroidal-momentum MS-HGNN on the Solo C2 graph (2 base nodes) -- drop-in for hgnn_c2_com.py:COM_HGNN_C2."""
    kind = "c2_com"
    num_bases = 2


class COM_HGNN_S4(_MSHGNNBase):
    """Centroidal-momentum baseline on the single-base graph (no masks / base MLP / residual) -- drop-in for
    hgnn_s4_com.py:COM_HGNN_S4.  symmetry_mode / group_operator_path are accepted and unused, as in the reference."""
    kind = "s4_com"
    num_bases = 1

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), symmetry_mode: str = None, group_operator_path: str = None):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.num_legs = 4
        self.num_joints = 12
        self.num_dimensions_per_base = 6
        self._build_convs(mean_rels=())
        self.decoder = pnn.Linear(hidden_channels, self.num_dimensions_per_base)


class COM_HGNN(_MSHGNNBase):
    """MI-HGNN with the decoder on the base node -- drop-in for hgnn.py:COM_HGNN (hgnn.py:66-117)."""
    kind = "s4_com"
    num_bases = 1

    def __init__(self, hidden_channels: int, num_layers: int, data_metadata, regression: bool = True,
                 activation_fn=nn.ReLU(), com_dimension: int = 6):
        super().__init__()
        self._init_common(hidden_channels, num_layers, data_metadata, regression, activation_fn)
        self.num_dimensions_per_base = com_dimension
        self._build_convs(mean_rels=())
        self.decoder = pnn.Linear(hidden_channels, self.num_dimensions_per_base)
