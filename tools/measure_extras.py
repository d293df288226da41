"""Side measurements quoted in DESIGN.md: window assembly rate, other configurations of the step."""
import json, os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphsym_hgnn_amd.windows import SequenceStore, quadsdk_a1_c2_recipe
N, T, B = 200_000, 150, 8192
rng = np.random.default_rng(0)
seq = {"imu_acc": rng.normal(size=(N, 3)), "imu_omega": rng.normal(size=(N, 3)), "q": rng.normal(size=(N, 12)), "qd": rng.normal(size=(N, 12)),
       "tau": rng.normal(size=(N, 12)), "F": rng.normal(size=(N, 12)), "r_o": rng.normal(size=(N, 4))}
store = SequenceStore(seq, quadsdk_a1_c2_recipe(range(12), range(4), T, 3), dtype="bf16")
starts = torch.randint(0, N - T, (B,)).cuda()
for _ in range(3): store.assemble(starts, reuse_buffers=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): store.assemble(starts, reuse_buffers=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gpu_ms = []
for _ in range(20):
    e0.record(); store.assemble(starts, reuse_buffers=True); e1.record(); torch.cuda.synchronize(); gpu_ms.append(e0.elapsed_time(e1))
gms = sorted(gpu_ms)[len(gpu_ms) // 2]
print(json.dumps({"window_assembly_ms_B8192_wall": dt * 1e3, "gpu_ms_events": gms, "windows_per_s_gpu": B / (gms * 1e-3), "GB_per_s_written_gpu": B * 14408 / (gms * 1e-3) / 1e9}))
for cfg in (["--layers", "8"], ["--batch", "32768"], ["--dtype", "f32"], ["--batch", "2048"]):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "15", "--warmup", "3", "--no-cpu-baseline"] + cfg, capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps({"cfg": cfg, "windows_per_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_us": d["kernel_us"]}))
