#!/bin/bash
# bench every kernel-experiment build in morphsym_hgnn_amd/csrc/variants/*.so (made with `make OUT=variants/x.so BUILD=build_x EXTRA=-D...`) next to the product library
ARGS="${@:---dtype x3}"
run() { python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['kernel_us'], 'loss', d['loss'])"; }
run product
for f in morphsym_hgnn_amd/csrc/variants/*.so; do MSHGNN_LIB=$PWD/$f run $(basename $f .so); done
