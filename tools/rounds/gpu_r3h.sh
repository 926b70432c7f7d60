#!/bin/bash
TAG=${1:-r3h}
mkdir -p gpurun_out
for v in 0 1 0 1; do echo "--- ANTMMF_ATTN_VARIANT=$v"; ANTMMF_ATTN_VARIANT=$v timeout 300 python tools/attn_bench.py v$v 10 2>&1 | grep "fwd.N257\|bwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_variants.txt
python -m pytest tests/test_hf_bert_bridge.py tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "bridge or bert or univl_arch or fold or towers" 2>&1 | tail -4
