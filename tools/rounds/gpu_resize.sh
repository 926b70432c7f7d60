#!/bin/bash
# resize parity tests + micro-benchmark + rocprofv3 per-kernel stats.  usage: gpu_resize.sh TAG
TAG=${1:-resize}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_kernels_gpu.py -q -k resize 2>&1 | tail -2
timeout 200 python tools/resize_bench.py 64 | tee gpurun_out/${TAG}_bench.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o prof -- python $GRAFT_REPO_ROOT/tools/resize_bench.py 64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|resize" "$f" | cut -c1-200 | tee gpurun_out/${TAG}_kernel_stats.csv
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -delete 2>/dev/null
