#!/bin/bash
# round 4: tail round of the persistent NT kernel split along K: 4 = product (split on), 33554436 = split off; hipBLASLt beside it; full-size GEMM tests; step A/B
TAG=${1:-r4x}
mkdir -p gpurun_out; export TMPDIR=/tmp
GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=4,33554436 GEMM_BENCH_NO_TN=1 timeout 400 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r4x_raw.log"):
    try: d = json.loads(l)
    except Exception:
        if not l.startswith("{"): print(l.strip()[:200])
        continue
    if "hipblaslt_tf_med" in d: print(d["shape"], "hipblaslt", d["hipblaslt_tf_med"], "own", d["own_tf_med"], d["own_over_hipblaslt"])
    elif "variant" in d: print(d["shape"], d["variant"], d["tf_med"], d["maxdiff_vs_v0"])
PY
cp gpurun_out/${TAG}_raw.log gpurun_out/${TAG}_gemm_tail_split_ab.jsonl
echo "=== gemm tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 900 -k "gemm" 2>&1 | tail -4 | cut -c1-400
echo "=== bench (split on)"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err | cut -c1-300; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4x_bench.json").read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
echo "=== bench (split off)"; ANTMMF_GEMM_VARIANT=33554436 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_off.json 2> gpurun_out/${TAG}_bench_off.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4x_bench_off.json").read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
