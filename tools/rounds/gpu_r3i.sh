#!/bin/bash
# bias-on-wgrad: parity of the changed kernels + whole GPU suite, wgrad micro A/B, default bench + rocprof stats.
TAG=${1:-r3i}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
python - <<'PY' 2>&1 | tee gpurun_out/${TAG}_wgrad_bias_ab.txt
import sys, torch, json
sys.path.insert(0, "ant-multi-modal-framework_amd")
from antmmf.hip import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
T = 263168
for n_out, k_in, tag in ((1024, 1024, "q/k/v/out"), (4096, 1024, "fc1"), (1024, 4096, "fc2"), (3072, 1024, "packed qkv")):
    dY = torch.randn(T, n_out, device=dev).to(BF); X = torch.randn(T, k_in, device=dev).to(BF)
    dW = torch.zeros(n_out, k_in, device=dev); db = torch.zeros(n_out, device=dev)
    r = {}
    for _ in range(3):
        r.setdefault("wgrad", []).append(timed(lambda: ops.gemm_wgrad_(dW, dY, X)))
        r.setdefault("wgrad+bias", []).append(timed(lambda: ops.gemm_wgrad_(dW, dY, X, db=db)))
        r.setdefault("colsum", []).append(timed(lambda: ops.colsum_(db, dY)))
    print(json.dumps({"shape": tag, "n_out": n_out, "k_in": k_in, **{k: round(sorted(v)[1], 4) for k, v in r.items()}}))
PY
echo "=== bench default"
timeout 900 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-600 gpurun_out/${TAG}_bench_l14.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14 -o prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_l14.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof_l14 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_l14_kernel_stats.csv && head -22 "$f" | cut -c1-150
find gpurun_out/${TAG}_prof_l14 -type f ! -name "*stats*" -delete 2>/dev/null
