#!/bin/bash
# needs an instrumented build: make -C morphsym_hgnn_amd/csrc clean && make -C morphsym_hgnn_amd/csrc EXTRA=-DMSHGNN_ABLATE (the product build ignores MSHGNN_DBG*)
# ablations of k_stack_fwd: 2 no MACs, 8 no epilogue, 64 no base_transform chain, 16 no X stash stores, 32 no relu bits
for d in 0 2 8 10 64 16 32 48 112; do
  MSHGNN_DBG=$d python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('dbg=$d', 'stack_fwd',k['stack_fwd'],'stack_bwd',k['stack_bwd'])"
done
