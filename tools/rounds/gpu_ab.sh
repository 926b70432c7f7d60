#!/bin/bash
# parity + bench A/B for an env knob.  usage: gpu_ab.sh TAG "ENV=VAL" 
TAG=${1:-ab}; KNOB=${2:-ANTMMF_GEMM_PERSIST=0}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench default"
timeout 900 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table.txt > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-1500 gpurun_out/${TAG}_bench.json
head -24 gpurun_out/${TAG}_gemm_table.txt
echo "=== bench with $KNOB"
env $KNOB timeout 900 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_knob.txt > gpurun_out/${TAG}_bench_knob.json 2>/dev/null; cut -c1-1500 gpurun_out/${TAG}_bench_knob.json
head -24 gpurun_out/${TAG}_gemm_table_knob.txt
