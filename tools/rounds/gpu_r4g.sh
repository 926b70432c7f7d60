#!/bin/bash
# round 4: attention forward on 32 x 32 MFMA tiles, A/B against the 16 x 16 kernel (ANTMMF_ATTN_VARIANT bit 0 = old kernel, bit 1 = 64-key softmax blocks)
TAG=${1:-r4g}
mkdir -p gpurun_out
for v in 1 0 2 1 0 2; do echo "--- ANTMMF_ATTN_VARIANT=$v"; ANTMMF_ATTN_VARIANT=$v timeout 300 python tools/attn_bench.py v$v 10 2>&1 | grep "fwd.N" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_fwd32.txt
python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -k "attention" 2>&1 | tail -3
