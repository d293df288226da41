#!/bin/bash
# needs an instrumented build: make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_ABLATE (the product build ignores MSHGNN_DBG*)
# finer epilogue ablations of k_layer_fwd: 16 no X_out global stores, 32 no relu-bit stores, 64 no base_transform chain, 128 no hb/t1 stores
for d in 0 16 32 48 64 128 112 240 8; do
  MSHGNN_DBG=$d python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('dbg=$d', 'fwd0',k['layer_fwd0'],'fwd1',k['layer_fwd1'],'fwd2',k['layer_fwd2'])"
done
