"""How much operand re-fetching the weight-gradient kernel's XCD placement leaves (host only, no GPU): compiles tools/probe/gradw_sharing.cpp
against the plan compiler and prints unique operand bytes per window vs the sum over XCDs of the distinct streams their lanes touch."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from morphsym_hgnn_amd import engine as eng

so = "/tmp/gradw_sharing.so"
subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/probe/gradw_sharing.cpp")] + sys.argv[3:], check=True)
lib = C.CDLL(so)
cfg = sys.argv[1] if len(sys.argv) > 1 else "a1c2"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
spec = bench.build_spec({"a1c2": 3, "mck4": 8}[cfg], cfg)
h = eng._DescHolder(spec, eng.DTYPE_CODES[dtype])
out = (C.c_double * 16)()
dump = (C.c_int32 * 40000)()
n = lib.probe(C.byref(h.desc), out, dump, 40000)
u, tot = out[0], out[1]
print(f"{cfg} {dtype}: lanes {int(out[2])} (padded {int(out[3])}), parts {int(out[4])}, items/lane {int(out[5])}")
print(f"unique operand bytes/window {u:.0f}; sum over XCDs {tot:.0f}; duplication x{tot / u:.3f}")
print("per XCD bytes/window:", [int(out[6 + x]) for x in range(8)])
if os.environ.get("DUMP"):
    a = np.array(dump[:n]).reshape(-1, 8)
    for x in range(8):
        print("XCD", x, [(int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[6])) for r in a if r[0] == x])
