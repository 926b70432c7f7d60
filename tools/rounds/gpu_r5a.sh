#!/bin/bash
# round 5, first call: the new real-width parity cases (report only: the gates are set from these numbers), the new e2e cases, a baseline bench of the round-start kernels
TAG=${1:-r5a}
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in l14 vtp8 dmae12; do
  echo "=== real width $c"; ANTMMF_REAL_WIDTH_REPORT_ONLY=1 timeout 900 python tests/real_width_case.py $c cuda:0 2>&1 | grep -v Warning | tail -4 | tee -a gpurun_out/${TAG}_real_width.log | cut -c1-1800
done
echo "=== new e2e cases"; timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "cnvid or temporal" 2>&1 | tail -4
echo "=== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-600 gpurun_out/${TAG}_bench_l14.json
