#!/bin/bash
for r in 1 9; do echo "== raster $r"; ANTMMF_GEMM_RASTER=$r timeout 200 python tools/kernel_bench.py 2>/dev/null | grep -E "gemm.wgrad" | cut -c1-120; done
