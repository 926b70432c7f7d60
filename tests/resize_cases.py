"""Shared cases for the resize kernel (CPU lane emulator and MI355X): product path vs the oracle / Pillow goldens, bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resize as oresize  # noqa: E402  (tests only)


def case_goldens(dev, golden):
    """Every golden case alone (uint8 output must equal Pillow's bytes), then all 3-channel cases with a common target as ONE ragged batch."""
    from antmmf.hip.image import resize_bicubic_u8

    g = golden("resize_bicubic.pt")
    for i, (h, w, c, oh, ow) in enumerate(g["cases"]):
        got = resize_bicubic_u8([g[f"in{i}"].to(dev)], oh, ow, out_f32=False)[0].cpu()
        assert torch.equal(got, g[f"out{i}"]), (i, (h, w, c, oh, ow), int((got.int() - g[f"out{i}"].int()).abs().max()))
    idx = [i for i, cs in enumerate(g["cases"]) if cs[2] == 3 and cs[3:] == (32, 32)]
    assert len(idx) >= 5
    got = resize_bicubic_u8([g[f"in{i}"].to(dev) for i in idx], 32, 32, out_f32=False).cpu()
    for j, i in enumerate(idx):
        assert torch.equal(got[j], g[f"out{i}"]), ("ragged batch", i)
    f = resize_bicubic_u8([g[f"in{i}"].to(dev) for i in idx], 32, 32, out_f32=True).cpu()
    want = torch.stack([g[f"out{i}"] for i in idx]).permute(0, 3, 1, 2).float().div(255)  # ToTensor
    assert torch.equal(f, want)
    return f"{len(g['cases'])} golden cases + ragged batch of {len(idx)}: byte-identical to Pillow {g['pillow_version']}"


def case_vs_oracle(dev, sizes, out_hw, seed=0, channels=3):
    """Seeded images of the given (h, w) sizes as one ragged batch vs the oracle (float ToTensor output, exact)."""
    from antmmf.hip.image import resize_bicubic_u8

    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 256, (h, w, channels), dtype=np.uint8) for h, w in sizes]
    got = resize_bicubic_u8([torch.from_numpy(i).to(dev) for i in imgs], out_hw, out_hw, out_f32=True).cpu().numpy()
    for j, im in enumerate(imgs):
        want = oresize.square_transform(im, out_hw)
        assert np.array_equal(got[j], want), (sizes[j], float(np.abs(got[j] - want).max()))
    return f"{len(sizes)} images x {channels} channels -> {out_hw}: equal to the oracle"
