"""Summary of tools/pmc_sq.sh's counter passes (gpurun_out/pmc_sq/{sq,tcc}/**/*counter_collection.csv): per kernel, per launch averages -- the share of wave
cycles spent waiting / waiting for an instruction / executing, MFMA busy cycles, L2 hit rate.  usage: summarize_sq.py gpurun_out/pmc_sq [bench_line.json]"""
import csv, glob, json, sys, collections
sys.path.insert(0, "tools")
from summarize_pmc import short      # noqa: E402  (kernel name -> stat name; summarize_pmc reads argv only under __main__ semantics below)


def per_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


root = sys.argv[1]
us = {}
if len(sys.argv) > 2:
    us = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]).get("kernel_us", {})
out = {}
for src in ("sq", "tcc"):
    for k, d in per_kernel(f"{root}/{src}/**/*counter_collection.csv").items():
        s = short(k)
        if s:
            out.setdefault(s, {}).update(d)
for s, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0.0) / wc
        d["wait_inst_any_frac"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
        d["active_inst_any_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    if d.get("TCC_REQ_sum"):
        d["l2_hit_rate"] = d.get("TCC_HIT_sum", 0.0) / (d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0))
    if s in us:
        d["avg_us_unperturbed"] = us[s]
        d["mfma_busy_frac"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * us[s] * 1e-6 * 2.4e9), 4)
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 2 --no-extras` (tools/pmc_sq.sh), per launch averages; SQ_* in quad-cycles except "
                   "SQ_VALU_MFMA_BUSY_CYCLES (cycles); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration at 2.4 GHz, duration from the "
                   "unperturbed bench line)", "kernels": out}, sys.stdout, indent=1)
