#!/usr/bin/env python
"""Where the persistent NT GEMM's time goes: in-kernel cycle stamps (build with -DANTMMF_GEMM_PROF -> lib/libantmmf_hip_prof.so)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("ANTMMF_PROF_LIB") or os.path.join(ROOT, "ant-multi-modal-framework_amd", "lib", "libantmmf_hip_prof.so")
os.environ["ANTMMF_HIP_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
import torch
from antmmf.hip import ops, _lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = _lib.load()
lib.antmmf_debug_gemm_prof.argtypes = [ctypes.c_void_p]
tokens = 257 * 1024
def run(name, n, k, bias, res):
    x = torch.randn(tokens, k, device=dev).to(BF); w = torch.randn(n, k, device=dev).to(BF) * 0.02
    b = torch.zeros(n, device=dev) if bias else None
    r = torch.randn(tokens, n, device=dev).to(BF) if res else None
    for _ in range(3): y = ops.gemm(x, w, bias=b, residual=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.gemm(x, w, bias=b, residual=r); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert lib.antmmf_debug_gemm_prof(buf) == 0
    v = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
    tot = v[:, :4].sum(1)
    frac = (v[:, :4].sum(0) / tot.sum()).tolist()
    print(json.dumps(dict(raster=os.environ.get("ANTMMF_GEMM_RASTER", "1"), case=name, ms=round(ms, 3), tflops=round(2.0 * tokens * n * k / ms / 1e9, 1), tiles_per_wg=v[:, 4].mean().item(),
                          frac_kloop=round(frac[0], 3), frac_operands=round(frac[1], 3), frac_prologue=round(frac[2], 3), frac_store=round(frac[3], 3),
                          cycles_per_tile=round((tot.sum() / v[:, 4].sum()).item()), stamped_over_elapsed=round(tot.mean().item() / (ms * 1e-3) / 1e9, 3))), flush=True)
run("fc1 (J=4096,R=1024) +bias", 4096, 1024, True, False)
run("dgrad-like (J=4096,R=1024) plain", 4096, 1024, False, False)
run("fc2 (J=1024,R=4096) +bias+res", 1024, 4096, True, True)
run("out-proj (J=1024,R=1024) +bias+res", 1024, 1024, True, True)
run("qkv (J=3072,R=1024) +bias", 3072, 1024, True, False)
