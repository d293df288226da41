#!/bin/bash
# needs an instrumented build: make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_ABLATE (the product build ignores MSHGNN_DBG*)
for d in 0 16 32 64 2 4 6 22 54 118; do
  MSHGNN_DBG=$d python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('dbg=$d', 'bwd0',k['layer_bwd0'],'bwd1',k['layer_bwd1'],'bwd2',k['layer_bwd2'])"
done
