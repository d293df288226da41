"""`torch.optim.Adam` for a model whose parameters live in the engine's flat buffer: the whole update is ONE launch.

`configure_optimizers` of the reference's wrappers returns `optim.Adam(self.parameters(), lr)` (gnnLightning.py:258-265).  With this
package's models the parameters are views of one flat fp32 buffer and their `.grad`s are views of one flat gradient buffer
(`models._MSHGNNBase._flat_params`, `_deliver_gradients`), so the ~50 per-tensor (or multi-tensor) kernels of torch's Adam collapse into
`mshgnn_adam_step` on the two flat buffers -- same arithmetic (torch defaults: no weight decay, no amsgrad; pinned against
`torch.optim.Adam` in tests/test_engine_gpu.py and tests/test_wrappers.py).

`FlatAdam` IS a `torch.optim.Adam` (same constructor defaults, `param_groups`, `state_dict()` / `load_state_dict()` with torch's own
per-parameter layout, `zero_grad`); whenever the flat layout is not in place at `step()` -- parameters moved or re-created, gradients
that are not the flat views (DDP buckets, a parameter without gradient), weight decay / amsgrad / maximize switched on, more than one
parameter group -- it carries its state over and runs torch's own step.
"""
from __future__ import annotations

import torch

from . import engine as eng


class FlatAdam(torch.optim.Adam):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, graph_safe: bool = False, **kw):
        """graph_safe: the step count lives on the device (`mshgnn_adam_step_counted`), so `step()` can be captured in a HIP graph and replayed
        (wrappers.GraphedTrainingStep) -- the flat route's counterpart of torch.optim.Adam(capturable=True)."""
        super().__init__(model.parameters(), lr=lr, betas=betas, eps=eps, **kw)
        self._model = model
        self._graph_safe = bool(graph_safe)
        self._t_dev = None                # graph_safe: device int64 step count (the truth; self._t is refreshed from it when the state is published)
        self._m = self._v = None          # flat exp_avg / exp_avg_sq (the per-parameter state entries are views of them)
        self._t = 0                       # steps taken on the flat route since the state was last synchronised with self.state
        self._owner = None                # the flat parameter buffer the flat state belongs to

    # ---- which route --------------------------------------------------------------------------------------------------
    def _flat_route(self):
        m = self._model
        if len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("capturable", False) \
                or g.get("differentiable", False) or isinstance(g["lr"], torch.Tensor):
            return None
        flat, gflat, gviews, params = getattr(m, "_flat", None), getattr(m, "_gflat", None), getattr(m, "_gviews", None), getattr(m, "_param_list", None)
        if not getattr(m, "_flat_ok", False) or flat is None or gflat is None or gviews is None or params is None or not flat.is_cuda:
            return None
        if len(params) != len(g["params"]) or any(a is not b for a, b in zip(params, g["params"])):
            return None
        if any(p.grad is not v for p, v in zip(params, gviews)):
            return None
        return flat, gflat, params, g

    # ---- state: flat buffers <-> torch's per-parameter entries -------------------------------------------------------------
    def _offsets(self):
        return list(self._model._spec.param_offsets().values())

    def _adopt(self, flat, params):
        """Start (or re-start, after the model re-created its flat buffer) the flat state from whatever torch-layout state exists."""
        self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        self._t, self._owner = 0, flat
        steps = set()
        for (o, n), p in zip(self._offsets(), params):
            st = self.state.get(p)
            if st:
                self._m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise RuntimeError("FlatAdam: the parameters' step counts differ; use torch.optim.Adam for this state")
        self._t = steps.pop() if steps else 0
        if self._graph_safe:
            self._t_dev = torch.tensor([self._t], dtype=torch.int64, device=flat.device)
        self._publish(params)

    def _publish(self, params):
        """torch's per-parameter state entries as VIEWS of the flat buffers (so state_dict() and a later torch-route step see them)."""
        for (o, n), p in zip(self._offsets(), params):
            self.state[p] = {"step": torch.tensor(float(self._t)), "exp_avg": self._m[o:o + n].view(p.shape), "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def _sync_steps(self):
        if self._graph_safe and self._t_dev is not None and self._owner is not None:
            self._t = int(self._t_dev.item())      # (a host sync: state_dict / route changes only, never inside a captured step)
        if self._owner is not None:
            for st in self.state.values():
                if "step" in st:
                    st["step"] = torch.tensor(float(self._t))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._owner = None                # re-adopt from the loaded per-parameter state at the next step

    # ---- step ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        route = self._flat_route()
        if route is None:
            self._sync_steps()
            self._owner = None               # (the per-parameter entries stay views of the old flat state; a return to the flat route re-adopts them)
            super().step()
            return loss
        flat, gflat, params, g = route
        if self._owner is not flat:
            self._adopt(flat, params)
        lib = eng.load_library()
        if self._graph_safe:
            with torch.cuda.device(flat.device):
                eng._check(lib, lib.mshgnn_adam_step_counted(flat.data_ptr(), gflat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), flat.numel(),
                                                             self._t_dev.data_ptr(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), 1.0,
                                                             torch.cuda.current_stream(flat.device).cuda_stream), "mshgnn_adam_step_counted")
            return loss
        self._t += 1
        with torch.cuda.device(flat.device):
            eng._check(lib, lib.mshgnn_adam_step(flat.data_ptr(), gflat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), flat.numel(), self._t,
                                                 float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), 1.0,
                                                 torch.cuda.current_stream(flat.device).cuda_stream), "mshgnn_adam_step")
        return loss
