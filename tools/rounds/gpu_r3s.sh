#!/bin/bash
for r in 1 2; do
  ANTMMF_HIP_LIB=$GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_base.so timeout 300 python tools/ln_bench.py base 2>&1 | grep "ln_fwd.*4096.act=gelu" | cut -c1-150
  timeout 300 python tools/ln_bench.py poly 2>&1 | grep "ln_fwd.*4096.act=gelu" | cut -c1-150
done | tee gpurun_out/r3s_ln_wave_poly.txt
