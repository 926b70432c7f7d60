#!/bin/bash
# Round-3 baseline of the committed code: parity suite, default bench (+ GEMM table), rocprofv3 stats of the default command, dmae12 (TPM-CL native) bench + stats.
TAG=${1:-r3f}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench default"
timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-1800 gpurun_out/${TAG}_bench_l14.json
echo "=== rocprofv3 --kernel-trace --stats of the default bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14 -o prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof_l14 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_l14_kernel_stats.csv && head -30 "$f" | cut -c1-170
find gpurun_out/${TAG}_prof_l14 -type f ! -name "*stats*" -delete 2>/dev/null
for wl in ${WORKLOADS:-dmae12}; do
  echo "=== bench $wl"
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_$wl.txt > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; tail -2 gpurun_out/${TAG}_bench_$wl.err; cut -c1-900 gpurun_out/${TAG}_bench_$wl.json
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$wl -o prof -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$wl.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/${TAG}_prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_${wl}_kernel_stats.csv && head -24 "$f" | cut -c1-170
  find gpurun_out/${TAG}_prof_$wl -type f ! -name "*stats*" -delete 2>/dev/null
done
