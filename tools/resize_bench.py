#!/usr/bin/env python
"""Micro-benchmark of the device resize (8(f4)): a batch of full-HD frames -> 224 x 224 ToTensor output; HIP events around the two
launches, algorithmic bytes = h*w*C (read) + 2*h*out_w*C (intermediate) + out; beside it Pillow on one host core."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
import numpy as np, torch
from antmmf.hip.image import ResizePlan, resize_bicubic_u8, resize_packed_u8
dev = torch.device("cuda:0")
n, h, w, s = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1080, 1920, 224
rng = np.random.default_rng(0)
host = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for _ in range(n)]
imgs = [t.to(dev) for t in host]
plan = ResizePlan([(h, w)] * n, 3, s, s, dev)
src = torch.cat([t.reshape(-1) for t in imgs]); tmp = torch.empty(plan.tmp_bytes, dtype=torch.uint8, device=dev)
for _ in range(2): out = resize_packed_u8(src, plan, True, tmp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = 5
e0.record()
for _ in range(it): out = resize_packed_u8(src, plan, True, tmp)   # the two kernels only
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
assert torch.equal(out, resize_bicubic_u8(imgs, s, s))
t0 = time.perf_counter(); out2 = resize_bicubic_u8(host, s, s); torch.cuda.synchronize(); ms_h2d = (time.perf_counter() - t0) * 1e3
alg = n * (h * w * 3 + 2 * h * s * 3 + s * s * 3 * 4)
res = dict(kernel="resize_h_u8 + resize_v_u8", images=n, in_hw=[h, w], out=s, ms=round(ms, 3), images_per_s=round(n / ms * 1e3),
           algorithmic_GBps=round(alg / ms / 1e6, 1), frac_of_8TBps=round(alg / ms / 1e6 / 8000, 4), ms_from_host_memory=round(ms_h2d, 2))
try:
    from PIL import Image
    t0 = time.perf_counter()
    for t in host[:8]: np.asarray(Image.fromarray(t.numpy()).resize((s, s), Image.BICUBIC))
    res["pillow_1core_images_per_s"] = round(8 / (time.perf_counter() - t0), 1)
except ImportError:
    pass
print(json.dumps(res))
