"""ms/step and us per 1000 windows of `bench.py --batch B` for a list of batch sizes (power-of-two node strides against padded ones).
usage: python tools/batch_sweep.py "8192 8208 8176" [bench.py arguments]"""
import json, subprocess, sys
for b in sys.argv[1].split():
    p = subprocess.run([sys.executable, "bench.py", "--batch", b, "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-extras"] + sys.argv[2:], capture_output=True, text=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print("B", b, "ms", round(d["ms_per_step"], 4), "us/kwin", round(d["ms_per_step"] * 1e6 / int(b), 3), {k: round(v, 1) for k, v in d["kernel_us"].items()}, flush=True)
