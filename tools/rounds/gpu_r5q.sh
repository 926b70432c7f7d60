#!/bin/bash
# round 5: one-kernel attention backward (persistent form everywhere) at the other towers' shapes (197 tokens x 12 heads: ViT-B/16) vs the two-kernel backward + the attention tests
TAG=${1:-r5q}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LAB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -k "attention or dropout" 2>&1 | tail -5
export ATTN_BENCH_SHAPES="${ATTN_BENCH_SHAPES:-1024x16x257,1024x12x197,512x12x197,1024x16x77,1024x12x77}"
for v in 8 0 8 0; do echo "--- ANTMMF_ATTN_VARIANT=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_VARIANT=$v timeout 300 python tools/attn_bench.py v$v 10 2>&1 | grep "bwd" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_other_shapes_ab.txt
