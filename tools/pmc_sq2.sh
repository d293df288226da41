#!/bin/bash
# instruction-mix counter passes over the bench; output gpurun_out/pmc_sq2/*.csv
out=gpurun_out/pmc_sq2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/a -o a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/a.log
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/b -o b -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/b.log
tail -n 2 $out/a.log $out/b.log
python3 - <<'P'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_sq2/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not (k.startswith("_Z") or k.startswith("k_")): continue
        k = k.split("(")[0][:24]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / max(1, cnt[(k, c)]) / 1e6, 2) for c, v in d.items()})
P
