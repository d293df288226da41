"""sha1 of the flat gradient (and loss bits) of one seeded step per configuration: run under two builds of the library (MSHGNN_LIB) to compare them bit for bit.
usage: python tools/grad_hash.py"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
for config, L, dtype, B in (("a1c2", 3, "bf16", 8192), ("a1c2", 3, "bf16", 48), ("a1c2", 3, "bf16", 50), ("a1c2", 3, "x3", 1040), ("a1c2", 3, "f32", 64), ("mck4", 8, "bf16", 4096), ("solo", 8, "bf16", 8192)):
    spec = bench.build_spec(L, config)
    e = eng.Engine(spec, dtype)
    flat = eng.flatten_params(spec, synth.make_params(3, spec.param_shapes()), e.device)
    x, y = bench.make_batch(spec, B, 17)
    xs = e.cast_inputs(x)
    if spec.regression:
        out, loss, g = e.step_mse(xs, flat, y.to(e.device, torch.float32).reshape(-1), B)
    else:
        out, loss, g = e.step_ce(xs, flat, y.to(e.device, torch.int32).reshape(B, -1).contiguous(), B)
    torch.cuda.synchronize()
    print(config, L, dtype, B, "loss", float(loss), "grad sha1", hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()[:16], "nonzero", int((g != 0).sum()))
