#!/bin/bash
# Round-end measurement set: parity, default bench (+cpu baseline, +gemm table), rocprofv3 stats of the default command,
# PMC traffic passes.  usage: gpu_final.sh TAG
TAG=${1:-final}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -3 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench default"
timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table.txt > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; tail -2 gpurun_out/${TAG}_bench_full.err; cat gpurun_out/${TAG}_bench_full.json
echo "=== rocprofv3 --kernel-trace --stats of the default bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_default -o prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -28 "$f" | cut -c1-170
find gpurun_out/${TAG}_prof_default -type f ! -name "*stats*" -delete 2>/dev/null
grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/${TAG}_prof_default.log | head -3
echo "=== PMC traffic"
bash tools/gpu_traffic.sh ${TAG} 1024
