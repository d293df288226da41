"""Where the host time of the nn.Module surface goes: cProfile over the step closures of bench.module_surface (fp64 inputs, plan-dtype inputs, the wrapper's
training_step x 2, the cast pass -- in that order), host-only cost per step (no synchronise inside the loop) next to the synchronised time.
usage: python tools/profile_module_surface.py [n_steps] [top]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 22
names = iter(["module fp64 inputs", "module plan-dtype inputs", "wrapper plan-dtype", "wrapper fp64", "module cast pass"])


def profiled(step, sync, steps, warmup):
    name = next(names)
    for _ in range(20):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t_enq = time.perf_counter() - t0
    sync()
    t_all = time.perf_counter() - t0
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        step()
    pr.disable()
    sync()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(TOP)
    print(f"=== {name}: enqueue {t_enq / N * 1e6:.1f} us/step on the host, {t_all / N * 1e6:.1f} us/step synchronised at the end", flush=True)
    if name in os.environ.get("PROFILE_CASES", "module plan-dtype inputs,wrapper plan-dtype").split(","):
        print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:TOP + 12]), flush=True)
    return t_all / N


bench.median_step_s = profiled
spec = bench.build_spec(3, "a1c2", 128)
dev = torch.device("cuda:0")
bench.module_surface(spec, 8192, dev, 30, 5, sys.argv[3] if len(sys.argv) > 3 else "bf16")
