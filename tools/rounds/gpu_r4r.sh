#!/bin/bash
# NOTE: the timing-only variants used here exist only in a library built with `make -C ant-multi-modal-framework_amd/csrc ABLATIONS=1` (rebuild the product library afterwards)
# round 4: does the SHAPE of a store instruction matter?  2105348 = rolling epilogue whose stores write 8 rows x 128 B (full cache lines) instead of 16 rows x 64 B (timing only: data misplaced)
TAG=${1:-r4r}
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
GEMM_BENCH_VARIANTS=8196,2105348,40964 GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
grep -E "dgrad" gpurun_out/${TAG}_raw.log | tee -a gpurun_out/${TAG}_gemm_full_line_stores.jsonl | cut -c1-230
grep -v "^{" gpurun_out/${TAG}_raw.log | tail -3
done
