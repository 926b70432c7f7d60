#!/bin/bash
# round 4: 32-byte store pieces in the rolling epilogue (two lane-bit exchanges in front of the stores): 4 = product dispatch (new layout), 4194308 = product dispatch with the old
# store layout (16 rows x 16 B per 16-lane pass), 16388 = burst epilogue, 8196 / 4202500 = rolling forced everywhere new / old
TAG=${1:-r4s}
mkdir -p gpurun_out; export TMPDIR=/tmp
GEMM_BENCH_VARIANTS=4,4194308,16388 GEMM_BENCH_NO_TN=1 timeout 400 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_raw.log 2>&1; echo "rc=$?"
grep "^{" gpurun_out/${TAG}_raw.log | tee gpurun_out/${TAG}_gemm_full_line_ab.jsonl | cut -c1-230
grep -v "^{" gpurun_out/${TAG}_raw.log | tail -3
echo "=== gemm / fold / e2e tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q --timeout 900 -k "gemm or ffn or fold or m2 or stage1 or dmae" 2>&1 | tail -4 | cut -c1-300
echo "=== bench (new layout)"; timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s_bench.json").read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
echo "=== bench (old layout)"; ANTMMF_GEMM_VARIANT=4194308 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_old.json 2> gpurun_out/${TAG}_bench_old.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s_bench_old.json").read().strip().splitlines()[-1])
print({k: d["roofline"][k] for k in ("achieved", "frac", "gemm_ms_per_step", "gemm_clock_mhz")}, d["value"], d["ms_per_step"])
PY
