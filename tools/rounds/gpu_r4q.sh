#!/bin/bash
# round 4: attention backward with -lse / -D as the accumulators' start values + padded key tile skipped; full GPU suite; step bench
TAG=${1:-r4q}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== attention bench"; for i in 1 2; do timeout 300 python tools/attn_bench.py new 10 2>&1 | grep "N257\|N77" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bench.jsonl
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-330 gpurun_out/${TAG}_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/%s_bench.json" % __import__("os").environ.get("TAG", "r4q")).read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
