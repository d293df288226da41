"""What the gradient SAMPLE check of tests/helpers.check_against_fixture really sees: per golden case and plan, the largest sampled-entry
error in units of (a) the reference gradient's rms (the unit the check uses, bound 30 x rtol) and (b) the largest sampled reference entry (a lower bound of the
tensor's max-abs, the unit every other stage of the harness uses)."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
worst = {}
for f in sorted(glob.glob(os.path.join(os.path.dirname(helpers.__file__), "golden", "*_B*.npz"))):
    name = os.path.basename(f)[:-4]
    if name.startswith("windows"):
        continue
    case, spec, fx, x_dict, y, params, ei = helpers.load_case(name)
    for dtype in ("f32", "x3"):
        try:
            errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, ei, case["B"], dtype=dtype, device="cuda:0")
        except Exception as ex:
            print(name, dtype, "skipped:", str(ex)[:80]); continue
        a_max = b_max = 0.0; who = None
        for key in fx.files:
            if not key.startswith("gnorm:"): continue
            k = key[6:]; g = grads[k].double().cpu().flatten().numpy(); gn = float(fx[key])
            if gn == 0.0: continue
            idx = helpers.sample_indices(k, g.size); ref = fx["gsample:" + k]
            e = float(np.abs(g[idx] - ref).max())
            a = e / max(gn / np.sqrt(g.size), 1e-30); b = e / max(float(np.abs(ref).max()), 1e-30)
            if a > a_max: a_max, who = a, k
            b_max = max(b_max, b)
        print(f"{name:32s} {dtype}: sample err / rms {a_max:.2e} ({who})   / max|sampled ref| {b_max:.2e}", flush=True)
        worst[dtype] = (max(worst.get(dtype, (0, 0))[0], a_max), max(worst.get(dtype, (0, 0))[1], b_max))
print("worst over all cases:", worst)
