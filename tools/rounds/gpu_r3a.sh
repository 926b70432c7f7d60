#!/bin/bash
# Round-3 first GPU pass: parity suite (with the new R2 / full-size video / RCCL tests), hipBLASLt A/B on the step's NT shapes, default bench.
TAG=${1:-r3a}
mkdir -p gpurun_out; export TMPDIR=/tmp
rocm-smi --showmeminfo vram 2>/dev/null | head -8; python -c "import torch; print('gpus', torch.cuda.device_count())"
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -15 gpurun_out/${TAG}_pytest_gpu.log
echo "=== hipBLASLt A/B (same process, same buffers, interleaved rounds)"
GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=4 timeout 600 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_gemm_bench_hipblaslt_1024pairs.jsonl 2>&1; cat gpurun_out/${TAG}_gemm_bench_hipblaslt_1024pairs.jsonl | cut -c1-400
echo "=== bench default"
timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -3 gpurun_out/${TAG}_bench_l14.err; cut -c1-1500 gpurun_out/${TAG}_bench_l14.json
