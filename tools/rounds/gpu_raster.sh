#!/bin/bash
for r in 0 1 2; do echo "== raster $r"; ANTMMF_GEMM_RASTER=$r timeout 200 python tools/kernel_bench.py 2>/dev/null | grep -E "gemm.fwd" | cut -c1-120; done
./tools/gemm_ablate | tail -12
