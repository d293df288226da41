"""Kernel text bytes of the built library, per __global__ symbol: the device code objects are pulled out of libmshgnn.so's fat binary
section with clang-offload-bundler and their symbol tables read with llvm-readelf.  usage: python tools/kernel_text_report.py [libmshgnn.so]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "morphsym_hgnn_amd", "libmshgnn.so")
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
raw = os.path.join(tmp, "fat.bin")
subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, raw], check=True)
data = open(raw, "rb").read()
# the section holds one clang offload bundle per translation unit, each starting with the magic string
starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
rows = []
for i, st in enumerate(starts):
    blob = os.path.join(tmp, f"b{i}.bin")
    open(blob, "wb").write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
    out = os.path.join(tmp, f"co{i}.o")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={blob}", f"--output={out}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"],
                       capture_output=True, text=True)
    if r.returncode or not os.path.exists(out):
        continue
    sym = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", out], capture_output=True, text=True).stdout
    for line in sym.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[4] == "GLOBAL" and not f[7].endswith(".kd"):
            rows.append((int(f[2]), f[7]))
rows = sorted(set(rows), reverse=True)      # (a symbol appears in .dynsym and .symtab)
total = sum(r[0] for r in rows)
print(f"{len(rows)} kernels, {total / 1024:.0f} KB of kernel text in {os.path.basename(lib)}")
for size, name in rows:
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{size:9d}  {dn[:150]}")
