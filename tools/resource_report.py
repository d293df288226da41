"""Per-kernel register / scratch report from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/resource_report.py [--all] /tmp/mshgnn.res ...   (files = stderr of `hipcc ... -S --cuda-device-only -Rpass-analysis=kernel-resource-usage`)"""
import re, subprocess, sys
show_all = "--all" in sys.argv
for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
    txt = open(path).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        sc, v, a, sp, ss, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        if show_all or sc > 0 or sp > 0 or ss > 0:
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{dn[:100]:100s} scratch {sc:4d}  vgpr {v:3d} agpr {a:3d}  vgpr-spill {sp:3d} sgpr-spill {ss:3d}  occ {occ}  lds {lds}")
