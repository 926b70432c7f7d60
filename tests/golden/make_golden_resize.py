#!/usr/bin/env python
"""Golden vectors for the image-resize step: seeded uint8 images and what Pillow's
`Image.resize((w, h), Image.BICUBIC)` -- the arithmetic behind torchvision Resize in the reference's `square_transform`
(prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14) -- returns for them.  Run in the build container (Pillow 12.2.0):
    python tests/golden/make_golden_resize.py  ->  tests/golden/resize_bicubic.pt
Cases: down / up-scaling, one pass skipped (w == out or h == out), both skipped, 1-pixel extents, grayscale, a ragged batch."""
import os

import numpy as np
import PIL
import torch
from PIL import Image

CASES = [  # (h, w, c, out_h, out_w)
    (37, 50, 3, 32, 32), (61, 45, 3, 32, 32), (32, 70, 3, 32, 32), (90, 32, 3, 32, 32), (32, 32, 3, 32, 32), (9, 11, 3, 32, 32),
    (1, 1, 3, 8, 8), (3, 200, 3, 16, 16), (200, 2, 3, 16, 16), (40, 33, 1, 24, 20), (130, 97, 3, 48, 56),
]


def main():
    rng = np.random.default_rng(20260927)
    out = {"pillow_version": PIL.__version__, "cases": CASES}
    for i, (h, w, c, oh, ow) in enumerate(CASES):
        img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        if i % 3 == 0:  # smooth content too (random noise alone never exercises the clamp asymmetrically)
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(255 * (np.sin(yy / 3.0 + k) * np.cos(xx / 5.0) > 0)).astype(np.uint8) for k in range(c)], axis=2)
        pil = Image.fromarray(img if c == 3 else img[:, :, 0], mode="RGB" if c == 3 else "L")
        ref = np.asarray(pil.resize((ow, oh), Image.BICUBIC))
        if c == 1:
            ref = ref[:, :, None]
        out[f"in{i}"] = torch.from_numpy(img.copy())
        out[f"out{i}"] = torch.from_numpy(ref.copy())
    torch.save(out, os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_bicubic.pt"))
    print("wrote resize_bicubic.pt with", len(CASES), "cases; Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
