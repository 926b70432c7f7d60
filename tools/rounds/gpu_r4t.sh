#!/bin/bash
# NOTE: the timing-only variants used here exist only in a library built with `make -C ant-multi-modal-framework_amd/csrc ABLATIONS=1` (rebuild the product library afterwards)
# round 4: store-shape ablations (timing only, data misplaced, NO exchange code): per 16-lane pass 16 rows x 16 B (4202500 = the old layout), 8 rows x 32 B (8396804),
# 4 rows x 64 B (16785412), 2 rows x 128 B (2105348); 8196 = the real thing (five exchanges, 2 rows x 128 B); 40964 = no stores
TAG=${1:-r4t}
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
GEMM_BENCH_VARIANTS=4202500,8396804,16785412,2105348,8196,40964 GEMM_BENCH_NO_TN=1 timeout 400 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
grep "dgrad" gpurun_out/${TAG}_raw.log | tee -a gpurun_out/${TAG}_gemm_store_shapes.jsonl | cut -c1-200
grep -v "^{" gpurun_out/${TAG}_raw.log | tail -3
done
