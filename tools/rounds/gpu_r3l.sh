#!/bin/bash
TAG=${1:-r3l}
mkdir -p gpurun_out
ALT=$GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_attn512.so
for r in 1 2; do
  timeout 300 python tools/attn_bench.py w6 10 2>&1 | grep "N257\|N77" | cut -c1-160
  ANTMMF_HIP_LIB=$ALT timeout 300 python tools/attn_bench.py w8 10 2>&1 | grep "N257\|N77" | cut -c1-160
done | tee gpurun_out/${TAG}_attn_waves.txt
ANTMMF_HIP_LIB=$ALT python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -x -k "attention" 2>&1 | tail -3
