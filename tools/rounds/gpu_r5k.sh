#!/bin/bash
# round 5: the one-kernel attention backward (attn_bwd_fused64_kernel) -- tests, same-box A/B against the two-kernel backward (lab variant bit 3), per-kernel time, the l14 step
TAG=${1:-r5k}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LAB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -k "attention or dropout" 2>&1 | tail -5
for v in 8 0 8 0; do echo "--- ANTMMF_ATTN_VARIANT=$v"; ANTMMF_HIP_LIB=$LAB ANTMMF_ATTN_VARIANT=$v timeout 300 python tools/attn_bench.py v$v 10 2>&1 | grep "bwd.N257\|fwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o p -- python $ROOT/tools/attn_bench.py prof 4 > $ROOT/gpurun_out/${TAG}_prof.log 2>&1)
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_attn_kernel_stats.csv && head -8 $f | cut -c1-200
echo "=== bench l14"; timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -1 gpurun_out/${TAG}_bench_l14.err; cut -c1-260 gpurun_out/${TAG}_bench_l14.json
