#!/bin/bash
TAG=${1:-r3j}
mkdir -p gpurun_out
for r in 1 2; do
  ANTMMF_HIP_LIB=$GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_base.so ANTMMF_ALLOW_EMULATOR=1 timeout 300 python tools/ln_bench.py base 2>&1 | grep "act=gelu\|copy" | cut -c1-220
  timeout 300 python tools/ln_bench.py poly 2>&1 | grep "act=gelu\|copy" | cut -c1-220
done | tee gpurun_out/${TAG}_ln_gelu_poly_ab.jsonl
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "layernorm or m2" 2>&1 | tail -3
