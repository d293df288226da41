#!/bin/bash
# two workgroups per CU: start the second half of the grid late (MSHGNN_STAGGER cycles); tools/stagger_sweep.sh "bench args" v1 v2 ...
ARGS="$1"; shift
for pass in 1 2; do for v in "$@"; do MSHGNN_STAGGER=$v python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger=$v', round(d['ms_per_step'],4), d['kernel_us'])"; done; done
