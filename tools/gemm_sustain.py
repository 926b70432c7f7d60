"""Sustained wall-clock throughput and effective shader clock of the product GEMM loop (VERDICT r5 next #2a: judge K-loop forms on wall-clock under the power cap).

    ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so python tools/gemm_sustain.py [seconds] [tag]

Back-to-back launches of one shape for >= `seconds` (default 10), per variant of the lab library: 4 = the product kernels, 4 | 32768 = the same loop without its
global stores (timing-only; comparable with tools/gemm_ablate sustain, whose loops store nothing).  One JSON line per ~1/4 s window (TF, clock of the last launch) and
one summary per run; A B A B so that drift of the box shows as a difference between the rounds.
"""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import _lib, ops  # noqa: E402


def clock_mhz(lib):
    buf = (ctypes.c_ulonglong * 2)()
    if lib.antmmf_debug_gemm_clock(buf) != 0 or buf[1] == 0:
        return None
    return round(buf[0] / (buf[1] / 100.0), 0)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    tag = sys.argv[2] if len(sys.argv) > 2 else "r6"
    lib = _lib.load()
    assert _lib.is_lab(), "needs the lab library (ANTMMF_HIP_LIB)"
    lib.antmmf_debug_gemm_clock.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    T = 257 * 1024
    shapes = {"fc1 263168x4096x1024 +bias": (T, 4096, 1024, True), "dgrad_fc1 263168x1024x4096 plain": (T, 1024, 4096, False)}
    out = []
    for name, (I, J, R, has_bias) in shapes.items():
        X = torch.randn(I, R, device=dev).bfloat16()
        W = (torch.randn(J, R, device=dev) * 0.03).bfloat16()
        b = torch.randn(J, device=dev) if has_bias else None
        Y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
        flop = 2.0 * I * J * R
        for rnd in range(2):
            for variant, vname in ((4, "product"), (4 | 32768, "no_stores(timing-only)")):
                lib.antmmf_debug_set_gemm_variant(variant)
                for _ in range(3):
                    ops.gemm(X, W, bias=b, out=Y)
                torch.cuda.synchronize()
                total, n, win = 0.0, 0, 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                while total < secs * 1e3:
                    e0.record()
                    for _ in range(40):
                        ops.gemm(X, W, bias=b, out=Y)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    total += ms
                    n += 40
                    if win % 4 == 0:
                        d = {"shape": name, "variant": vname, "round": rnd, "t_s": round(total * 1e-3, 2), "tflops": round(flop * 40 / ms / 1e9, 1), "clock_mhz": clock_mhz(lib)}
                        print(json.dumps(d), flush=True)
                        out.append(d)
                    win += 1
                d = {"shape": name, "variant": vname, "round": rnd, "sustained_s": round(total * 1e-3, 1), "launches": n, "tflops_avg": round(flop * n / total / 1e9, 1)}
                print(json.dumps(d), flush=True)
                out.append(d)
        del X, W, Y
    lib.antmmf_debug_set_gemm_variant(4)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/{tag}_gemm_sustain_product.jsonl", "w") as f:
        for d in out:
            f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
