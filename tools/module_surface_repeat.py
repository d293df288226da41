"""bench.module_surface with every route measured twice in a row (is the first figure of a route an ordering artefact?).  usage: python tools/module_surface_repeat.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

orig = bench.median_step_s
names = iter([f"{n}, pass {i}" for n in ("module fp64", "module plan-dtype", "wrapper plan-dtype", "wrapper fp64", "module cast pass") for i in (1, 2)])      # (bench takes the better of two passes per route)


def twice(step, sync, steps, warmup, *a, **k):
    n = next(names)
    r = [orig(step, sync, steps, warmup, *a, **k) for _ in range(2)]
    print(n, [round(x * 1e3, 4) for x in r], flush=True)
    return r[-1]


bench.median_step_s = twice
bench.module_surface(bench.build_spec(3, "a1c2", 128), 8192, torch.device("cuda:0"), 30, 5, "bf16")
