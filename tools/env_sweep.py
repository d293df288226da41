"""`bench.py` under a list of values of one environment variable ("-" = unset).  usage: python tools/env_sweep.py VAR "v1 v2 -" [bench.py arguments]"""
import json, os, subprocess, sys
var = sys.argv[1]
for v in sys.argv[2].split():
    env = dict(os.environ)
    env.pop(var, None)
    if v != "-":
        env[var] = v
    p = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--min-time", "0.2"] + sys.argv[3:], capture_output=True, text=True, env=env)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print(var, v, "ms", round(d["ms_per_step"], 4), {k: round(x, 1) for k, x in d["kernel_us"].items()}, flush=True)
