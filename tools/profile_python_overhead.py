"""Host-side cost of one training step of examples/train_flat.py (the GPU work is asynchronous)."""
import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from examples import train_flat
pr = cProfile.Profile()
pr.enable()
train_flat.train(steps=300, batch=8192, dtype="bf16", log_every=10_000, quiet=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
