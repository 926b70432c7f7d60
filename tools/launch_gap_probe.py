"""What one more launch costs a stream: N back-to-back launches of a kernel that does (almost) nothing, wall-clock per launch.

    python tools/launch_gap_probe.py [N]

`cast` = this library's fp32 -> bf16 cast on 64 elements through the C ABI (ctypes), `torch_add` = torch's x.add_(1) on 64 elements, `copy` = a 2-MB device-to-device copy
(the size of one packed q / k / v weight).  Host-side enqueue time is printed next to the stream time: the stream time is what the step pays when the host runs ahead."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    x = torch.zeros(64, device=dev)
    out = torch.empty(64, device=dev, dtype=torch.bfloat16)
    a, b = torch.zeros(1 << 20, device=dev, dtype=torch.bfloat16), torch.zeros(1 << 20, device=dev, dtype=torch.bfloat16)
    for name, fn in (("cast", lambda: ops.cast_bf16(x, out=out)), ("torch_add", lambda: x.add_(1.0)), ("copy_2MB", lambda: b.copy_(a))):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(json.dumps({"kernel": name, "launches": n, "us_per_launch_on_the_stream": round(e0.elapsed_time(e1) * 1e3 / n, 2), "us_per_launch_host_enqueue": round(t_host * 1e6 / n, 2)}), flush=True)


if __name__ == "__main__":
    main()
