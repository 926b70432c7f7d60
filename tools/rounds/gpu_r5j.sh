#!/bin/bash
# round 5: the gated dgrad (out = acc * gate) on the rolling-epilogue kernel -- GEMM tests, the e2e cases that run it, the video workloads' step + per-shape GEMM tables
TAG=${1:-r5j}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -k "gemm or stage1 or full_size or real_width_step_vs_oracle or real_width_vs_oracle" 2>&1 | tail -3
for wl in vtp8 dmae12 vtp8t; do
  echo "=== bench $wl"; timeout 600 python bench.py --workload $wl --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_$wl.txt > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; tail -1 gpurun_out/${TAG}_bench_$wl.err; cut -c1-200 gpurun_out/${TAG}_bench_$wl.json; head -4 gpurun_out/${TAG}_gemm_table_$wl.txt
done
