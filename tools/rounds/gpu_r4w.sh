#!/bin/bash
# round 4: the final NT / TN kernels against hipBLASLt on the step's shapes, same process and buffers (as profiles/r3_gemm_bench_vs_hipblaslt_1024pairs.jsonl)
TAG=${1:-r4w}
mkdir -p gpurun_out; export TMPDIR=/tmp
GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=4 timeout 600 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_gemm_bench_vs_hipblaslt_1024pairs.jsonl 2>&1; echo "rc=$?"
grep hipblaslt gpurun_out/${TAG}_gemm_bench_vs_hipblaslt_1024pairs.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d.get('shape'), d.get('epi'), d.get('hipblaslt_tf_med'), d.get('own_tf_med'), d.get('own_over_hipblaslt'))
"
