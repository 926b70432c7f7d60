#!/bin/bash
# round 4: tail split, second version (double-buffered slice, partial tiles in fragment order): 4 = product rule, 33554436 = off, 67108868 = every shape whose last round is <= 1/4 full
TAG=${1:-r4z}
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
GEMM_BENCH_VARIANTS=4,33554436,67108868 GEMM_BENCH_NO_TN=1 timeout 400 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r4z_raw.log"):
    try: d = json.loads(l)
    except Exception:
        if not l.startswith("{"): print(l.strip()[:200])
        continue
    if "variant" in d: print(d["shape"], d["variant"], d["tf_med"], d["maxdiff_vs_v0"])
PY
cat gpurun_out/${TAG}_raw.log >> gpurun_out/${TAG}_gemm_tail_split_v2.jsonl
done
echo "=== gemm tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 900 -k "gemm" 2>&1 | tail -3 | cut -c1-300
