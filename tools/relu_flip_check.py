"""Are the gradient outliers of a plan on tiny batches relu decisions that flip within the arithmetic error?  usage: python tools/relu_flip_check.py x3|f32
(finding, round 2: yes -- every outlier coincides with a decision whose pre-activation lies within 1e-4 of zero)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
from morphsym_hgnn_amd import synth, engine as eng
from oracle import ms_hgnn_oracle as orc
spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
dt = sys.argv[1]
for B, seed in [(17, 117), (17, 1), (17, 2), (17, 3), (65, 165), (65, 1), (16, 1), (16, 2), (33, 1), (33, 2)]:
    x_dict, y = synth.make_windows(seed, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(5, spec.param_shapes())
    cfg = helpers.oracle_config(spec)
    ei = spec.topology.edge_index_dict(B)
    o_out, o_hidden = orc.forward(cfg, params, {k: v.clone() for k, v in x_dict.items()}, ei, return_hidden=True)
    _, _, o_grads = orc.step(cfg, params, x_dict, ei, y, B)
    e = eng.Engine(spec, dt)
    xs = e.cast_inputs(x_dict)
    flat = eng.flatten_params(spec, params, e.device)
    out, loss, g = e.step_mse(xs, flat, y.reshape(-1).to(e.device, torch.float32), B)
    torch.cuda.synchronize()
    grads = eng.unflatten(spec, g.cpu())
    worst = max((float((grads[k].double() - r).abs().max() / r.abs().max()), k) for k, r in o_grads.items() if float(r.abs().max()) > 0)
    flips = 0
    sl = helpers.node_slices(spec)
    for l in range(spec.num_layers + 1):
        ref = helpers.dense_hidden(spec, o_hidden[l], B)
        got = e.hidden_state(B, l).double().cpu()
        if l == 0:
            ro, re_ = ref, got
        else:
            ro = ref - helpers.dense_hidden(spec, o_hidden[l - 1], B)
            re_ = got - e.hidden_state(B, l - 1).double().cpu()
        types = spec.node_types if l == 0 else [t for t in spec.live_types(l - 1) if t != "base"]
        for t in types:
            a, b = ro[:, sl[t]], re_[:, sl[t]]
            near = (a.abs() < 1e-4) | (b.abs() < 1e-4)
            mism = near & ((a > 0) != (b > 1e-9 * 0 + 0)) & ((a - b).abs() < 1e-4)
            flips += int(mism.sum())
    print(f"B={B} seed={seed} worst grad err {worst[0]:.2e} ({worst[1]})  relu decisions that differ within 1e-4 of zero: {flips}")
