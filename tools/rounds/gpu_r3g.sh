#!/bin/bash
# Sub-LN fold: parity (new tests first), per-op microbench, then the default bench with the fold on / off.
TAG=${1:-r3g}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_frames.py -m gpu -q --timeout 900 -x -k "ffn_fold or frames or processor or dark or full_size" > gpurun_out/${TAG}_pytest_new.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_new.log
python -m pytest tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "m2" > gpurun_out/${TAG}_pytest_m2.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_m2.log
timeout 600 python tools/ffn_fold_bench.py 1024 257 5 > gpurun_out/${TAG}_ffn_fold_bench_image.jsonl 2> gpurun_out/${TAG}_ffn_fold_bench.err; cat gpurun_out/${TAG}_ffn_fold_bench_image.jsonl; tail -3 gpurun_out/${TAG}_ffn_fold_bench.err
echo "=== bench fold on"
timeout 900 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_l14_fold.txt > gpurun_out/${TAG}_bench_l14_fold.json 2> gpurun_out/${TAG}_bench_l14_fold.err; tail -2 gpurun_out/${TAG}_bench_l14_fold.err; cut -c1-700 gpurun_out/${TAG}_bench_l14_fold.json
echo "=== bench fold off"
ANTMMF_FFN_FOLD=0 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_l14_nofold.json 2> gpurun_out/${TAG}_bench_l14_nofold.err; tail -2 gpurun_out/${TAG}_bench_l14_nofold.err; cut -c1-300 gpurun_out/${TAG}_bench_l14_nofold.json
