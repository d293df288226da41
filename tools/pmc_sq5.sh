#!/bin/bash
# Round 5: SQ / SQC counter passes of the step kernels: wave-cycle breakdown, instruction mix, instruction-cache and scalar-cache hit rates.
# usage (on the GPU box, from the repo root): bash tools/pmc_sq5.sh <tag> [bench.py arguments, e.g. --dtype x3]; output gpurun_out/pmc5_<tag>/{sq,mix,sqc}
tag=${1:-bf16}; shift
ARGS="$@"
out=gpurun_out/pmc5_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --min-time 0.01 $ARGS"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/sq -o sq -- $B > /dev/null 2> $out/sq.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/mix -o mix -- $B > /dev/null 2> $out/mix.log
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ --kernel-trace --output-format csv -d $out/sqc -o sqc -- $B > /dev/null 2> $out/sqc.log
tail -2 $out/sq.log $out/mix.log $out/sqc.log
ls $out/*
