#!/bin/bash
TAG=${1:-r3q}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "gemm or univl_stage1 or full_size or ffn_fold" 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "--- ANTMMF_GEMM_ACT16=$v"
  ANTMMF_GEMM_ACT16=$v timeout 600 python bench.py --workload vtp8 --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_vtp8_$v.txt 2>/dev/null | cut -c1-230 | sed 's/.*"value"/"value"/'
  grep "gelu " gpurun_out/${TAG}_gemm_table_vtp8_$v.txt | head -3
done | tee gpurun_out/${TAG}_act16_ab.txt
