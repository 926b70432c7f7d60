#!/bin/bash
# round 4: the restricted tail split (>= 16 slices of >= 4 K-tiles): GEMM tests incl. the full-size tail test, step A/B
TAG=${1:-r4y}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== gemm tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 900 -k "gemm" 2>&1 | tail -4 | cut -c1-400
for v in 4 33554436 4 33554436; do
echo "=== bench variant $v"; ANTMMF_GEMM_VARIANT=$v timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$v.json 2> gpurun_out/${TAG}_bench.err; V=$v python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/r4y_bench_%s.json" % os.environ["V"]).read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
done
