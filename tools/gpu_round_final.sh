#!/bin/bash
# The per-round measurement set (same sequence since round 2): parity on the product library (+ the real-width / contract reports), default bench (+cpu baseline, +gemm table),
# rocprofv3 stats of the default command, the other workloads with their own kernel stats, PMC traffic passes, own vs hipBLASLt per shape (lab library), loss kernels.
# usage: gpu_round_final.sh TAG [skip_traffic]
TAG=${1:-r6}
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/${TAG}_real_width.jsonl; rm -f $ANTMMF_REAL_WIDTH_OUT
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
unset ANTMMF_REAL_WIDTH_OUT
echo "=== kernel inventory of the product library"; python tools/kernel_inventory.py --json gpurun_out/${TAG}_kernel_inventory.json | tail -1
echo "=== bench default"
timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cat gpurun_out/${TAG}_bench_l14.json
echo "=== rocprofv3 --kernel-trace --stats of the default bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14 -o prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof_l14 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_l14_kernel_stats.csv && python tools/kernel_families.py "$f" | tee gpurun_out/${TAG}_bench_l14_families.txt
t=$(find gpurun_out/${TAG}_prof_l14 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/kernel_trace_split.py "$t" > gpurun_out/${TAG}_bench_l14_row_kernels_by_grid.txt
find gpurun_out/${TAG}_prof_l14 -type f ! -name "*stats*" -delete 2>/dev/null
for wl in b16 vtp8 vtp8t dmae12; do
  echo "=== bench $wl"
  timeout 600 python bench.py --workload $wl --gemm-table gpurun_out/${TAG}_gemm_table_$wl.txt > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; tail -2 gpurun_out/${TAG}_bench_$wl.err; cut -c1-900 gpurun_out/${TAG}_bench_$wl.json
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$wl -o prof -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$wl.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/${TAG}_prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_${wl}_kernel_stats.csv && head -14 "$f" | cut -c1-150
  find gpurun_out/${TAG}_prof_$wl -type f ! -name "*stats*" -delete 2>/dev/null
done
if [ -z "$2" ]; then echo "=== PMC traffic"; bash tools/gpu_traffic.sh ${TAG} 1024; fi
echo "=== loss kernels at the global-batch slab"; timeout 300 python tools/loss_bench.py 2>&1 | grep kernel | tee gpurun_out/${TAG}_loss_bench.jsonl
echo "=== own (lab library, default variant = the product's kernels) vs hipBLASLt, per shape"; GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=4 timeout 900 tools/gemm_bench 1024 3 2>&1 | tee gpurun_out/${TAG}_gemm_bench_vs_hipblaslt_1024pairs.jsonl | cut -c1-200
