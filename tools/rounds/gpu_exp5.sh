#!/bin/bash
python tools/gemm_exp3.py 2>/dev/null | head -6
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k gemm 2>&1 | tail -2
timeout 200 python tools/kernel_bench.py 2>/dev/null | grep -E "gemm" | cut -c1-120
