import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers
from morphsym_hgnn_amd import synth
spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
for B in [int(a) for a in sys.argv[1:]] or [16, 17, 33]:
    x_dict, y = synth.make_windows(100 + B, B, spec.num_nodes, spec.widths, 12)
    params = synth.make_params(5, spec.param_shapes())
    errs, out, loss, grads = helpers.run_engine_case(spec, x_dict, y, params, spec.topology.edge_index_dict(B), B, dtype="x3")
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print("B", B, [(k, f"{v:.2e}") for k, v in top])
