#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or wgrad" 2>&1 | tail -2
ANTMMF_GEMM_RASTER=49 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or wgrad" 2>&1 | tail -2
for r in 1 49 57; do ANTMMF_GEMM_RASTER=$r python tools/gemm_exp_tn.py; done 2>&1 | tee gpurun_out/exp_tn2.jsonl
