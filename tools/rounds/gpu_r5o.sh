#!/bin/bash
# round 5: the persistent one-kernel attention backward -- tests, same-box A/B against the two-kernel backward (lab variant bit 3), timing-only ablations (lab library):
#   ANTMMF_ATTN_FUSED_ABL = 1 no dQ contraction, 2 no per-chunk barrier, 4 no exponential, 8 no dV / dK contraction, 16 no O rows (D pass), 32 no 17th key tile, 15 = 1+2+4+8,
#   128 = K^T fragments NOT in registers (the A/B of that change)
TAG=${1:-r5o}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LAB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -k "attention or dropout" 2>&1 | tail -5
for v in 8 0 8 0; do echo "--- ANTMMF_ATTN_VARIANT=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_VARIANT=$v timeout 300 python tools/attn_bench.py v$v 10 2>&1 | grep "bwd.N257\|fwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_ab.txt
for v in 1 2 4 8 16 32 15 128 0; do echo "--- ANTMMF_ATTN_FUSED_ABL=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_FUSED_ABL=$v timeout 300 python tools/attn_bench.py abl$v 10 2>&1 | grep "bwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_ablations.txt
