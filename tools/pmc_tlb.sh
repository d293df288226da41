#!/bin/bash
# TLB (UTCL1 / UTCL2) counter passes over the bench; output gpurun_out/pmc_tlb/*.csv
out=gpurun_out/pmc_tlb; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L 2>/dev/null | grep -i -E "UTCL|TLB" | head -40 > $out/avail.txt
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $out/utcl1 -o u1 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/u1.log
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/lat -o lat -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/lat.log
tail -3 $out/u1.log $out/lat.log
python3 - <<'P'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_tlb/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:28]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / max(1, cnt[(k, c)])) for c, v in d.items()})
P
