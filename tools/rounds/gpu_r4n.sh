#!/bin/bash
# NOTE: the timing-only variants used here exist only in a library built with `make -C ant-multi-modal-framework_amd/csrc ABLATIONS=1` (rebuild the product library afterwards)
# round 4: store ablations of the rolling-epilogue GEMM, timing only (wrong values by construction), R = 1024 plain shapes:
#   270340   one 16 x 32 block stored per K-tile per wave instead of the whole tile at its end (stores spread evenly over the K loop)
#   532484   the same, but all DMA pieces issued by waves 4 - 7 (waves 0 - 3 store and never wait on vmcnt)
#   1056772  every wave issues its own DMA; all stores (two blocks per K-tile) by waves 0 - 3
#   1581060  both: DMA only by waves 4 - 7, stores only by waves 0 - 3 -- no wave has stores and DMA pieces in the same vmcnt counter
TAG=${1:-r4n}
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 16388,8196,40964,270340,1581060; do
echo "=== variants $v"
GEMM_BENCH_VARIANTS=$v GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
grep -E "dgrad_fc2|dgrad_out" gpurun_out/${TAG}_raw.log | cut -c1-260 | tee -a gpurun_out/${TAG}_gemm_store_ablation.jsonl | cut -c1-200
grep -v "^{" gpurun_out/${TAG}_raw.log | tail -3
done
