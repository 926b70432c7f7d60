#!/bin/bash
./tools/gemm_ablate | grep -E "32x32|full loop:|mfma only  " | head -12
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "layernorm or activations" 2>&1 | tail -2
timeout 200 python tools/kernel_bench.py 2>/dev/null | grep -E "layernorm|act" | cut -c1-120
