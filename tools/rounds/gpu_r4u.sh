#!/bin/bash
# round 4: attention rows stored as 16-byte pieces; attention + e2e tests; step
TAG=${1:-r4u}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== attention bench"; for i in 1 2; do timeout 300 python tools/attn_bench.py st16 10 2>&1 | grep "N257\|N77" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bench.jsonl
echo "=== tests"; timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_hf_bert_bridge.py -m gpu -x -q --timeout 900 2>&1 | tail -3 | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4u_bench.json").read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
