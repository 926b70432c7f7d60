#!/bin/bash
# Which kernels does hipBLASLt pick for the step's shapes (names encode the Tensile tile configuration), and what resources do they use?
TAG=${1:-r3b}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "full_size or registry" > gpurun_out/${TAG}_pytest_new.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_new.log
cd /tmp && GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=0 GEMM_BENCH_NO_TN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_lt -o prof -- $GRAFT_REPO_ROOT/tools/gemm_bench 1024 1 $GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip.so > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_lt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof_lt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_hipblaslt_kernel_stats.csv && cut -c1-700 "$f" | head -40
t=$(find gpurun_out/${TAG}_prof_lt -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then head -1 "$t"; python - "$t" <<'PY'
import csv, sys
seen = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if n not in seen:
        seen[n] = r
for n, r in seen.items():
    if "Cijk" in n or "gemm_nt" in n:
        print({k: r[k] for k in r if k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size")})
PY
fi
find gpurun_out/${TAG}_prof_lt -type f ! -name "*stats*" -delete 2>/dev/null
