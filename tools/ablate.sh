#!/bin/bash
# needs an instrumented build: make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_ABLATE (the product build ignores MSHGNN_DBG*)
# timing-only ablations of k_layer_fwd (results are wrong by construction; only kernel_us matters)
for d in 0 1 2 4 8 6 14 15; do
  MSHGNN_DBG=$d python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('dbg=$d', 'fwd0',k['layer_fwd0'],'fwd1',k['layer_fwd1'],'fwd2',k['layer_fwd2'])"
done
