#!/bin/bash
# SQ / TCC counter passes over the generic engine's bench (--config synth32): per-kernel wave-cycle breakdown, instruction mix, L2 hit rate
out=gpurun_out/pmc_gen; mkdir -p $out
ARGS="--config synth32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --min-time 0.01 $@"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq -o sq -- python3 bench.py $ARGS > /dev/null 2> $out/sq.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $out/mix -o mix -- python3 bench.py $ARGS > /dev/null 2> $out/mix.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $out/tcc -o tcc -- python3 bench.py $ARGS > /dev/null 2> $out/tcc.log
python3 - <<'P'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_gen/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_g" not in k: continue
        k = k.split("(")[0][-28:]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / max(1, cnt[(k, c)]) / 1e6, 2) for c, v in d.items()})
P
