"""8-wave stack kernels against the slab ones between one and two tiles per CU.  usage: python tools/slab_threshold_sweep.py "4112 4608 5120 5632 6128" [bench.py arguments]"""
import json, os, subprocess, sys
for b in sys.argv[1].split():
    row = {}
    for v in ("0", "2"):
        env = dict(os.environ, MSHGNN_SLAB=v)
        p = subprocess.run([sys.executable, "bench.py", "--batch", b, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--min-time", "0.2"] + sys.argv[2:], capture_output=True, text=True, env=env)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        row[v] = (round(d["ms_per_step"], 4), round(d["kernel_us"].get("stack_step", 0.0), 1))
    print("B", b, "tiles", (int(b) + 15) // 16, "8wave", row["0"], "slab", row["2"], flush=True)
