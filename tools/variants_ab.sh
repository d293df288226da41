#!/bin/bash
# A/B of kernel-experiment builds: tools/variants_ab.sh "bench args" name1 name2 ...  (names under morphsym_hgnn_amd/csrc/variants/, "product" = the shipped library); two alternating passes
ARGS="$1"; shift
run() { python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['kernel_us'], 'loss', d['loss'])"; }
for pass in 1 2; do
  for n in "$@"; do
    if [ "$n" = product ]; then run product; else MSHGNN_LIB=$PWD/morphsym_hgnn_amd/csrc/variants/$n.so run $n; fi
  done
done
