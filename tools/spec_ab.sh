#!/bin/bash
# specialised step kernel on / off, two alternating passes: tools/spec_ab.sh "bench args"
ARGS="$1"
run() { MSHGNN_SPEC=$1 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spec=$1', round(d['ms_per_step'],4), d['kernel_us'], 'loss', d['loss'])"; }
for pass in 1 2; do run 1; run 0; done
