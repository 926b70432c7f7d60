#!/bin/bash
TAG=${1:-r3m}
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -x -k "attention" 2>&1 | tail -3
for r in 1 2; do timeout 300 python tools/attn_bench.py tail 10 2>&1 | grep "N257\|N77" | cut -c1-160; done | tee gpurun_out/${TAG}_attn_tail.txt
