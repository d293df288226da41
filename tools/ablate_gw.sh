#!/bin/bash
# needs an instrumented build: make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_ABLATE (the product build ignores MSHGNN_DBG*)
for d in 0 1 2 4 8 3 7 15; do
  MSHGNN_DBG_GW=$d python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('dbg_gw=$d', 'gradw',k['gradw'])"
done
