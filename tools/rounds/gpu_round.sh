#!/bin/bash
# One gpurun call: parity tests, smoke, bench (small + full), rocprof kernel stats.  Logs -> gpurun_out/.
TAG=${1:-r1}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -15 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench small (l14, batch 64)"
timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_b64.json 2> gpurun_out/${TAG}_bench_b64.err; tail -3 gpurun_out/${TAG}_bench_b64.err; cat gpurun_out/${TAG}_bench_b64.json
echo "=== bench mid (l14, batch 256)"
timeout 600 python bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_b256.json 2> gpurun_out/${TAG}_bench_b256.err; tail -3 gpurun_out/${TAG}_bench_b256.err; cat gpurun_out/${TAG}_bench_b256.json
echo "=== bench full (l14, batch 1024) + cpu baseline"
timeout 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; tail -3 gpurun_out/${TAG}_bench_full.err; cat gpurun_out/${TAG}_bench_full.json
echo "=== rocprof kernel stats (batch 256, 2 steps)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/${TAG}_prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the stats csv (traces are large)
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -delete 2>/dev/null
