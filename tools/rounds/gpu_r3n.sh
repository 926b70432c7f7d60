#!/bin/bash
TAG=${1:-r3n}
mkdir -p gpurun_out
for r in 1 2; do
  timeout 300 python tools/attn_bench.py w8 10 2>&1 | grep "N77" | cut -c1-160
  for n in 256 320 384; do ANTMMF_HIP_LIB=$GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_attn$n.so timeout 300 python tools/attn_bench.py t$n 10 2>&1 | grep "N77" | cut -c1-160; done
done | tee gpurun_out/${TAG}_attn_text_threads.txt
