#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01d
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/write.log
python3 tools/summarize_pmc.py $out/fetch $out/write > $out/pmc_traffic.json
python3 bench.py --steps 30 --warmup 5 > $out/bench_line.json 2> $out/bench.log
ls -la $out $out/stats/* | head -40
