#!/bin/bash
# round 4: the sub-LN fold with fc1 on the rolling two-output epilogue: per-op bench, fold tests, step A/B (ANTMMF_FFN_FOLD=1)
TAG=${1:-r4m}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== ffn fold bench (image tokens)"; timeout 600 python tools/ffn_fold_bench.py 1024 257 5 2>&1 | tee gpurun_out/${TAG}_ffn_fold_bench_image.jsonl | cut -c1-200
echo "=== fold tests"; python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 900 -k "ffn_fold or fold" 2>&1 | tail -4
