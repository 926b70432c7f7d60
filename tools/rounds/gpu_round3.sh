#!/bin/bash
# parity + default bench (with per-shape GEMM table) + wgrad workgroup-count A/B + rocprof csv.  Logs -> gpurun_out/.
TAG=${1:-r1p}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench default"
timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table.txt > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; tail -3 gpurun_out/${TAG}_bench_full.err; cat gpurun_out/${TAG}_bench_full.json
head -40 gpurun_out/${TAG}_gemm_table.txt
echo "=== rocprof kernel stats (batch 256, 2 steps)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-200
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -delete 2>/dev/null
