"""The two-output activation forward of the CLIP / BERT feed-forwards (C = act(X W^T + b), aux = act'(.)) -- the video workloads' slowest GEMM -- with
TIMING-ONLY ablations of its second output (lab library, ANTMMF_GEMM_VARIANT bits 28 / 29): one byte per element instead of two (what an 8-bit stored derivative would write),
not stored at all (the single-output bound), next to the plain bias GEMM of the same shape.

    ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so python tools/gemm_two_output_bench.py [iters]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import _lib, ops  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    assert _lib.is_lab(), "needs the lab library (ANTMMF_HIP_LIB)"
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (T, J, R, act) in ((352256, 3072, 768, "gelu"), (100864, 3072, 768, "quick_gelu")):
        X = torch.randn(T, R, device=dev).bfloat16()
        W = (torch.randn(J, R, device=dev) * R ** -0.5).bfloat16()
        b = torch.randn(J, device=dev) * 0.1
        aux = torch.empty(T, J, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * T * J * R
        for rep in (1, 2):
            for name, variant, fn in (
                ("bias only (one output)", 4, lambda: ops.gemm(X, W, bias=b)),
                ("bias + act, two outputs (product: activation fixed at compile time)", 4, lambda: ops.gemm(X, W, bias=b, act=act, aux=aux, aux_grad=True)),
                ("bias + act, two outputs, run-time activation id (the form before)", 4 | (1 << 30), lambda: ops.gemm(X, W, bias=b, act=act, aux=aux, aux_grad=True)),
                ("... second output 1 B / element (timing only)", 4 | (1 << 28), lambda: ops.gemm(X, W, bias=b, act=act, aux=aux, aux_grad=True)),
                ("... second output not stored (timing only)", 4 | (1 << 29), lambda: ops.gemm(X, W, bias=b, act=act, aux=aux, aux_grad=True)),
            ):
                lib.antmmf_debug_set_gemm_variant(variant)
                ms = timeit(fn, iters)
                print(json.dumps({"shape": f"{T}x{J}x{R} {act}", "rep": rep, "form": name, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
        lib.antmmf_debug_set_gemm_variant(4)


if __name__ == "__main__":
    main()
