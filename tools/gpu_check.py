"""Diagnostic: run every golden case through the HIP engine and print per-stage relative errors vs the oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    names = sys.argv[2:] or helpers.GOLDEN_CASES
    worst_all = 0.0
    for name in names:
        case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
        try:
            errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype=dtype)
        except Exception as ex:  # noqa: BLE001
            print(f"{name}: EXCEPTION {type(ex).__name__}: {ex}")
            continue
        worst = max(errs.values())
        worst_all = max(worst_all, worst)
        print(f"== {name} [{dtype}] worst rel err {worst:.3e}")
        for k, v in errs.items():
            flag = "  <<<<" if v > (1e-4 if dtype == "f32" else 5e-2) else ""
            if flag or not k.startswith("grad:") or v > 1e-5:
                print(f"   {k:70s} {v:.3e}{flag}")
    print("WORST", worst_all)


if __name__ == "__main__":
    main()
