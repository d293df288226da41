import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
from morphsym_hgnn_amd import engine as eng
from oracle import ms_hgnn_oracle as orc
os.environ["MSHGNN_ENGINE"] = "generic"
case, spec, fx, x_dict, y, params, ei = helpers.load_case("mi_h128_L2_d1_B3")
B = case["B"]
# make the encoder trivially checkable: zero all encoder weights except a chosen K column
for dt in ("x3", "bf16"):
    for kcol in (0, 5, 8, 127, 128, 300):
        p2 = {k: v.clone() for k, v in params.items()}
        w = torch.zeros_like(p2["encoder.lins.joint.weight"]); w[:, kcol] = 1.0
        p2["encoder.lins.joint.weight"] = w
        p2["encoder.lins.joint.bias"] = torch.zeros_like(p2["encoder.lins.joint.bias"])
        e = eng.Engine(spec, dt)
        out = e.forward(e.cast_inputs(x_dict), eng.flatten_params(spec, p2, e.device), B, training=True)
        torch.cuda.synchronize()
        X0 = e.hidden_state(B, 0).double().cpu()           # [B, NN, h]
        sl = helpers.node_slices(spec)["joint"]
        ref = torch.relu(x_dict["joint"][:, kcol]).view(B, 12, 1).expand(B, 12, 128)
        got = X0[:, sl]
        print(dt, "kcol", kcol, "max err", float((got - ref).abs().max()), "ref max", float(ref.abs().max()), "got[0,0,:4]", got[0, 0, :4].tolist(), "ref", float(ref[0, 0, 0]))
