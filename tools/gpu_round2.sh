#!/bin/bash
# probe + parity + kernel bench + full bench + rocprof csv.  Logs -> gpurun_out/.
TAG=${1:-r1c}
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/probe_gfx950 > gpurun_out/${TAG}_probe.txt 2>&1; echo "probe rc=$?"
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python tools/kernel_bench.py > gpurun_out/${TAG}_kernel_bench.jsonl 2> gpurun_out/${TAG}_kernel_bench.err; grep -E "gemm|attention" gpurun_out/${TAG}_kernel_bench.jsonl | cut -c1-160
echo "=== bench default (l14, batch 1024) + bounded cpu baseline"
timeout 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; tail -3 gpurun_out/${TAG}_bench_full.err; cat gpurun_out/${TAG}_bench_full.json
echo "=== rocprof kernel stats (batch 256, 2 steps)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -delete 2>/dev/null
