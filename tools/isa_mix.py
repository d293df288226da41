"""Static instruction mix per kernel from hipcc's -S output: python tools/isa_mix.py file.s [name-filter]"""
import re, sys, collections
path = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; mix = {}
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m: cur = m.group(1); mix[cur] = collections.Counter(); continue
    if line.startswith('.Lfunc_end'): cur = None
    if cur and line.startswith('\t') and not line.startswith('\t.') and not line.startswith('\t;'):
        op = line.strip().split()[0]
        c = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "branch" if op.startswith("s_cbranch") or op.startswith("s_branch") else
             "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else "smem" if op.startswith("s_load") or op.startswith("s_buffer_load") else
             "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem_ld" if "load" in op else "vmem_st" if "store" in op else "other")
        mix[cur][c] += 1
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): mix[cur]["(v_mov)"] += 1
        if "readlane" in op or "readfirstlane" in op: mix[cur]["(readlane)"] += 1
for k, v in mix.items():
    if flt in k:
        tot = sum(c for n, c in v.items() if not n.startswith("("))
        print(k[:70], "total", tot, dict(sorted(v.items())))
