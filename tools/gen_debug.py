import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
for name, env in [("a1c2_h128_L1_d1_B2", "generic"), ("a1c2_h256_L2_d3_B3", ""), ("synth8_mi_h256_L3_B3", ""), ("mi_h128_L2_d1_B3", "generic")]:
    os.environ["MSHGNN_ENGINE"] = env
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="x3")
    print(name, {k: f"{v:.1e}" for k, v in errs.items() if not k.startswith("grad:")})
    bad = sorted(((v, k) for k, v in errs.items() if k.startswith("grad:") and v > 1e-4), reverse=True)[:6]
    print("   worst grads:", [(k, f"{v:.1e}") for v, k in bad])
