"""Summary of tools/pmc_sq5.sh's passes (gpurun_out/pmc5_<tag>/{sq,mix,sqc}/*counter_collection.csv): per kernel, per launch averages -- wave-cycle
breakdown, instruction mix, instruction-cache / scalar-cache hit rates.  usage: summarize_sq5.py gpurun_out/pmc5_<tag> [bench_line.json]"""
import csv, glob, json, sys, collections
sys.path.insert(0, "tools")
from summarize_pmc import short      # noqa: E402


def per_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        # one row per (dispatch, counter, dimension instance?) -- sum the instances of a dispatch, then average the dispatches
        disp = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            disp[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in disp.items():
            agg[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


root = sys.argv[1]
us = {}
if len(sys.argv) > 2:
    us = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]).get("kernel_us", {})
out = {}
for src in ("sq", "mix", "sqc"):
    for k, d in per_kernel(f"{root}/{src}/**/*counter_collection.csv").items():
        s = short(k)
        if s:
            out.setdefault(s, {}).update(d)
for s, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            d[c.lower()[3:] + "_frac"] = round(d.get(c, 0.0) / wc, 4)
    if d.get("SQC_ICACHE_REQ"):
        d["icache_hit_rate"] = round(d.get("SQC_ICACHE_HITS", 0.0) / d["SQC_ICACHE_REQ"], 4)
        d["icache_miss_rate"] = round((d.get("SQC_ICACHE_MISSES", 0.0) + d.get("SQC_ICACHE_MISSES_DUPLICATE", 0.0)) / d["SQC_ICACHE_REQ"], 4)
    if d.get("SQC_DCACHE_REQ"):
        d["dcache_hit_rate"] = round(d.get("SQC_DCACHE_HITS", 0.0) / d["SQC_DCACHE_REQ"], 4)
    if s in us:
        d["avg_us_unperturbed"] = us[s]
        d["mfma_busy_frac"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * us[s] * 1e-6 * 2.4e9), 4)
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 2 --no-extras` (tools/pmc_sq5.sh), per launch averages (instances of a dispatch summed); "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "(1024 SIMDs x unperturbed kernel duration at 2.4 GHz)", "kernels": out}, sys.stdout, indent=1)
