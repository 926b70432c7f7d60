#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "layernorm or m2" 2>&1 | tail -2
for r in 1 2; do timeout 300 python tools/ln_bench.py res 2>&1 | grep "ln_bwd" | cut -c1-150; done | tee gpurun_out/r3t_ln_bwd_resident.txt
