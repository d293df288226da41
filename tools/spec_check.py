"""Specialised step kernel against the interpreting one: same plan, same inputs -> every output, loss and gradient bit-identical (run on the GPU box).
usage: python tools/spec_check.py [layers] [windows] [config]"""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
CFG = sys.argv[3] if len(sys.argv) > 3 else "a1c2"
if os.environ.get("SPEC_CHILD"):
    import torch, bench
    from morphsym_hgnn_amd import engine as eng, synth
    dev = torch.device("cuda", 0)
    spec = bench.build_spec(L, CFG)
    e = eng.Engine(spec, "bf16", device=dev)
    x, y = bench.make_batch(spec, B, 0)
    xs = e.cast_inputs(x); flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
    yd = y.to(dev) if spec.regression else y.to(dev)
    res = e.step_mse(xs, flat, yd.float().reshape(-1), B) if spec.regression else e.step_ce(xs, flat, yd.reshape(-1).int(), B)
    torch.cuda.synchronize()
    out_t, loss, grad = res[0], res[1], res[2]
    np.save(os.environ["SPEC_CHILD"], np.concatenate([loss.float().cpu().numpy().reshape(-1), grad.float().cpu().numpy().reshape(-1), out_t.float().cpu().numpy().reshape(-1)]))
    sys.exit(0)
out = []
for spec_on in ("1", "0"):
    f = f"/tmp/spec_check_{spec_on}.npy"
    env = dict(os.environ, SPEC_CHILD=f, MSHGNN_SPEC=spec_on)
    subprocess.run([sys.executable, __file__, str(L), str(B), CFG], env=env, check=True)
    out.append(np.load(f))
a, b = out
same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
print(json.dumps({"config": CFG, "layers": L, "windows": B, "loss_spec": float(a[0]), "loss_interp": float(b[0]), "bit_identical": bool(same),
                  "max_abs_diff": float(np.abs(a - b).max()), "norm": float(np.linalg.norm(a[1:]))}))
sys.exit(0 if same else 1)
