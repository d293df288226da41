#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02a [bench.py arguments, e.g. --dtype x3]
tag=${1:-rXX}; shift
ARGS="$@"
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --min-time 0.05 $ARGS > $out/bench_under_rocprof.json 2> $out/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-time 0.01 $ARGS > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-time 0.01 $ARGS > /dev/null 2> $out/write.log
python3 bench.py --steps 30 --warmup 5 $ARGS > $out/bench_line.json 2> $out/bench.log
python3 tools/summarize_pmc.py $out/fetch $out/write $out/bench_line.json > $out/pmc_traffic.json
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
ls $out
