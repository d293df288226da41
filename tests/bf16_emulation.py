"""Test infrastructure: float64 model of the engine's bf16 plan -- the same algebra as the oracle, restructured the
way the kernels compute it (root weights pre-summed per destination type, per-edge accumulation), with a
round-to-bf16 at exactly the points where the bf16 plan stores or feeds bf16:
inputs, packed weights, X_l, base_transform H / T1, and (backward) every stored activation gradient.
With quant=False it must reproduce the oracle to ~1e-13 (tests/test_bf16_emulation.py), which validates the
restructuring; with quant=True it is what the bf16 HIP path must match up to fp32-accumulation effects."""
import torch

from morphsym_hgnn_amd.spec import rel_key


class _Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _mk(quant):
    q = (lambda x: _Q.apply(x)) if quant else (lambda x: x)                       # activations: value + gradient rounding
    qw = (lambda w: w + (w.to(torch.bfloat16).to(w.dtype) - w).detach()) if quant else (lambda w: w)  # weights: value only
    return q, qw


def emulate_step(spec, params, x_dict, y, B, quant=True, loss_grad=None):
    """Returns (out [B*n_out, d], loss, grads dict).  `loss_grad`: optional callable(out)->loss."""
    q, qw = _mk(quant)
    P = {k: v.detach().clone().double().requires_grad_(True) for k, v in params.items()}
    masks = spec.input_masks()
    nn_ = spec.num_nodes
    X = {}
    for t in spec.node_types:
        x = x_dict[t].double()
        if quant:
            x = x.to(torch.bfloat16).double()
        x = x.view(B, nn_[t], -1) * masks[t].unsqueeze(0)
        X[t] = q(torch.relu(x @ qw(P[f"encoder.lins.{t}.weight"]).t() + P[f"encoder.lins.{t}.bias"]))   # [B, n_t, h]
    for l in range(spec.num_layers):
        live = spec.live_types(l)
        new = {}
        for t in live:
            rels = [et for et in spec.edge_types if et[2] == t]
            w_root = sum(P[f"convs.{l}.convs.{rel_key(et)}.lin_root.weight"] for et in rels)
            b_sum = sum(P[f"convs.{l}.convs.{rel_key(et)}.lin_rel.bias"] for et in rels)
            H = X[t] @ qw(w_root).t() + b_sum
            for et in rels:
                w = qw(P[f"convs.{l}.convs.{rel_key(et)}.lin_rel.weight"])
                for (j, i) in spec.topology.edges(et):
                    upd = torch.zeros_like(H)
                    upd[:, i, :] = X[et[0]][:, j, :] @ w.t()
                    H = H + upd
            if spec.has_base_transform and t == "base":
                hb = q(H)
                t1 = q(torch.relu(hb @ qw(P["base_transform.0.weight"]).t() + P["base_transform.0.bias"]))
                Y = t1 @ qw(P["base_transform.2.weight"]).t() + P["base_transform.2.bias"]
            else:
                Y = torch.relu(H)
            new[t] = q(Y + X[t]) if spec.residual else q(Y)
        for t in spec.node_types:
            if t not in new:
                new[t] = X[t]   # dead: never read again
        X = new
    out = X[spec.out_type] @ P["decoder.weight"].t() + P["decoder.bias"]          # decoder keeps fp32 weights
    out = (out * spec.output_mask().unsqueeze(0)).reshape(B * nn_[spec.out_type], -1)
    if loss_grad is not None:
        loss = loss_grad(out)
    elif spec.regression:
        loss = ((out.reshape(-1) - y.double().reshape(-1)) ** 2).mean()
    else:
        loss = torch.nn.functional.cross_entropy(out.reshape(B * 4, 2), y.long().flatten())
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
    return out.detach(), loss.detach(), grads
