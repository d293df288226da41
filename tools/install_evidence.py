"""Copy the profile sets `bash tools/profile_round.sh <tag> ...` left under gpurun_out/prof_<tag>/ into profiles/ under their committed names and print what they hold.
usage: python tools/install_evidence.py <tag> [<tag> ...] [--lines file.json ...]   (files after --lines: bench lines under gpurun_out/ copied as they are)"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = (("k_prep", "prep"), ("k_enc", "enc"), ("k_slab_step", "stack_step"), ("k_stack_step", "stack_step"), ("k_stack_fwd", "stack_fwd"), ("k_stack_bwd", "stack_bwd"), ("k_gradw", "gradw"),
        ("k_finalize", "finalize"), ("k_gstep", "gstep"), ("k_ggradw", "ggradw"), ("k_gfinalize", "gfinalize"), ("k_gagg", "gagg"), ("k_gdec", "gdec"), ("k_dec_bwd", "dec_bwd"))


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for key, s in KEYS:
            if key in r["Name"]:
                o = out.setdefault(s, [0.0, 0]); o[0] += float(r["TotalDurationNs"]); o[1] += int(r["Calls"]); break
    return {k: round(v[0] / v[1] / 1e3, 2) for k, v in out.items()}


args, lines = sys.argv[1:], []
if "--lines" in args:
    i = args.index("--lines"); args, lines = args[:i], args[i + 1:]
for tag in args:
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    b = json.loads(open(f"{d}/bench_line.json").read().strip().splitlines()[-1])
    t = json.load(open(f"{d}/pmc_traffic.json"))
    real = {k: v for k, v in t["kernels"].items() if "alias_of" not in v}
    print(tag, "ms/step", round(b["ms_per_step"], 4), "commit", b.get("commit"), "source_hash", b["source_hash"], "==" if b["source_hash"] == t.get("source_hash") else "!=", t.get("source_hash"),
          "write_scale", t.get("write_scale"))
    print("   rocprofv3 avg us", stats(f"{d}/kernel_stats.csv"))
    print("   HBM-side MB", {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in real.items()}, "total", round(sum(v["hbm_bytes"] for v in real.values()) / 1e6))
    for f in ("kernel_stats.csv", "pmc_traffic.json", "bench_line.json", "bench_under_rocprof.json"):
        shutil.copy(f"{d}/{f}", os.path.join(ROOT, "profiles", f"{tag}_{f}"))
for f in lines:
    shutil.copy(os.path.join(ROOT, "gpurun_out", f), os.path.join(ROOT, "profiles", f))
    b = json.loads(open(os.path.join(ROOT, "gpurun_out", f)).read().strip().splitlines()[-1])
    print(f, "ms/step", round(b["ms_per_step"], 4), "source_hash", b.get("source_hash"), "traffic_source", b["roofline"].get("traffic_source"))
