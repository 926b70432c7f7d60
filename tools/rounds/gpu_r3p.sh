#!/bin/bash
TAG=${1:-r3p}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -x -k "movers or gemm or tpmcl or token_weight or pair" 2>&1 | tail -3
for wl in dmae12 vtp8; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_$wl.txt > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; tail -1 gpurun_out/${TAG}_bench_$wl.err; cut -c1-330 gpurun_out/${TAG}_bench_$wl.json
done
python - <<'PY'
import sys, torch
sys.path.insert(0, "ant-multi-modal-framework_amd")
from antmmf.hip import ops
dev = torch.device("cuda:0")
ids = torch.randint(0, 50000, (1024, 77), device=dev); dx = torch.randn(1024, 77, 1024, device=dev).to(torch.bfloat16); tab = torch.zeros(50000, 1024, device=dev)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
print("embed scatter sorted ms", timed(lambda: ops.embed_scatter_add_(tab, dx, ids)))
ops.SCATTER_SORT_MIN_ROWS = 1 << 60
print("embed scatter atomic ms", timed(lambda: ops.embed_scatter_add_(tab, dx, ids)))
print("pos scatter ms", timed(lambda: ops.embed_scatter_add_(tab, dx, None, None, seq=77, offset=0)))
PY
