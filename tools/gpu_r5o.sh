#!/bin/bash
# round 5: the persistent one-kernel attention backward -- tests, same-box A/B against the two-kernel backward, timing-only ablations
TAG=${1:-r5o}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LAB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
echo skip tests
echo skip ab
for v in 1 2 4 8 16 32 15 128 0; do echo "--- ANTMMF_ATTN_FUSED_ABL=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_FUSED_ABL=$v timeout 300 python tools/attn_bench.py abl$v 10 2>&1 | grep "bwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_ablations.txt
