#!/bin/bash
# round 4: the two-output activation epilogue on the rolling kernel: race screen + vtp8 / dmae12 step A/B against the burst epilogue (variant bit 14)
TAG=${1:-r4l}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== race screen"; timeout 600 python tools/gemm_race_screen.py 4 256 3 2>&1 | tee gpurun_out/${TAG}_race.jsonl | cut -c1-220 | tail -5
for wl in vtp8 dmae12; do
echo "=== $wl default"; timeout 600 python bench.py --workload $wl --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_$wl.txt > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; cut -c1-260 gpurun_out/${TAG}_bench_$wl.json; head -6 gpurun_out/${TAG}_gemm_table_$wl.txt
echo "=== $wl burst epilogue"; ANTMMF_GEMM_VARIANT=16388 timeout 600 python bench.py --workload $wl --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_${wl}_burst.txt > gpurun_out/${TAG}_bench_${wl}_burst.json 2>/dev/null; cut -c1-260 gpurun_out/${TAG}_bench_${wl}_burst.json; head -6 gpurun_out/${TAG}_gemm_table_${wl}_burst.txt
done
echo "=== loss bench"; timeout 300 python tools/loss_bench.py 2>&1 | tail -5
echo "=== loss tests at full size"; python -m pytest tests/test_loss_full_size_gpu.py tests/test_kernels_gpu.py -m gpu -q --timeout 900 -k "global_batch or gemm_k64 or softmax or milnce" 2>&1 | tail -5
