"""Library load, plan creation and first-step latency (18 MB of kernel text: is any of it paid at start-up?).  usage: python tools/startup_time.py"""
import time; t0=time.perf_counter()
import os, sys, torch; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morphsym_hgnn_amd import engine as eng, synth
t1=time.perf_counter()
spec=bench.build_spec(3); e=eng.Engine(spec,"bf16"); t2=time.perf_counter()
flat=eng.flatten_params(spec, synth.make_params(0, spec.param_shapes()), e.device)
x,y=bench.make_batch(spec,32,1); xs=e.cast_inputs(x)
out,loss,g=e.step_mse(xs,flat,y.to(e.device,torch.float32).reshape(-1),32); torch.cuda.synchronize(); t3=time.perf_counter()
out,loss,g=e.step_mse(xs,flat,y.to(e.device,torch.float32).reshape(-1),32); torch.cuda.synchronize(); t4=time.perf_counter()
print(f"imports {t1-t0:.2f}s  plan create {t2-t1:.2f}s  first step (incl. input prep) {t3-t2:.2f}s  second step {t4-t3:.4f}s")
