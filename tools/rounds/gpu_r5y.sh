#!/bin/bash
# round 5: A/B of lab variants of the 257-token one-kernel attention backward (ANTMMF_ATTN_FUSED_ABL: 256 = dQ contraction of chunk c - 1 in the barrier interval of chunk c's scores)
TAG=${1:-r5y}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LAB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
export ATTN_BENCH_SHAPES="1024x16x257"
for v in ${ABLS:-0 256 0 256}; do echo "--- ANTMMF_ATTN_FUSED_ABL=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_FUSED_ABL=$v timeout 300 python tools/attn_bench.py abl$v 10 2>&1 | grep "bwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_variants_ab.txt
