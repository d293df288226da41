import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
spec = bench.build_spec(3)
dev = torch.device("cuda", 0)
e = eng.Engine(spec, dtype="bf16", device=dev)
B = 8192
g = torch.Generator().manual_seed(1234)
imu = torch.randn(B, 1, 900, generator=g)
x = {"base": imu.expand(B, 2, 900).reshape(B * 2, 900), "joint": torch.randn(B * 12, 450, generator=g), "foot": torch.ones(B * 4, 1)}
xs = e.cast_inputs(x)
y = torch.randn(B * 12, generator=g).to(dev)
flat = eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), dev)
gflat = torch.empty_like(flat); out = torch.empty(B * 4, 3, dtype=torch.float32, device=dev); loss = torch.empty(1, device=dev)
def step():
    e.step_mse(xs, flat, y, B, out=out, grad_flat=gflat, loss=loss)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 50
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
gref = gflat.clone()
for _ in range(5): gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): gr.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 50
print("eager ms", t_eager * 1e3, "graph ms", t_graph * 1e3, "same grads", bool(torch.equal(gref, gflat)), float(loss))
