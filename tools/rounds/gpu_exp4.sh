#!/bin/bash
for r in 1 9 8; do echo "== raster $r"; ANTMMF_GEMM_RASTER=$r python tools/gemm_exp3.py 2>/dev/null | head -1; done
