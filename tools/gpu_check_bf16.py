"""Diagnostic for the bf16 plan: norm-wise relative errors (||a-b|| / ||b||) per gradient tensor."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers
from oracle import ms_hgnn_oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "a1c2_h128_L2_d3_B37"
case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype="bf16")
o_out, o_loss, o_grads = orc.step(helpers.oracle_config(spec), params, x_dict, ei, y, case["B"])
print("out maxrel", errs["out"])
rows = []
for k, g in o_grads.items():
    n = float(g.norm())
    if n == 0:
        rows.append((0.0, k, 0.0, float(grads[k].abs().max()))); continue
    e = float((grads[k].double() - g).norm()) / n
    rows.append((e, k, n, errs["grad:" + k]))
rows.sort(reverse=True)
for e, k, n, m in rows[:25]:
    print(f"{k:70s} l2rel {e:.3e}  maxrel {m:.3e}  |g| {n:.3e}")
