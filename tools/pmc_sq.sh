#!/bin/bash
# SQ / TCC counter passes over the bench (per-kernel wave-cycle breakdown and L2 hit rate); output gpurun_out/pmc_sq/*.csv
out=gpurun_out/pmc_sq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq -o sq -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $out/sq.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $out/tcc -o tcc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $out/tcc.log
tail -3 $out/sq.log $out/tcc.log
ls $out/*
