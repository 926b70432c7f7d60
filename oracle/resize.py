"""TEST INFRASTRUCTURE (oracle): CPU restatement of the image resize in front of the M2 image tower.

Path: `square_transform(size)` = torchvision `Resize((size, size), interpolation=BICUBIC)` + `ToTensor()`
(reference: prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14; caller prj/M2_Encoder/m2_encoder.py:61-68 and
vlmo/utils/beit_utils.py:72).  For a PIL image torchvision's Resize is `img.resize((w, h), BICUBIC)`, i.e. the arithmetic lives in
a third-party dependency that is not vendored in /root/reference: **Pillow** (requirements.txt:19 `pillow>=4.3.0`,
prj/M2_Encoder/requirements.txt:4 unpinned; 12.2.0 in this image).  Restated here from Pillow's published algorithm
(src/libImaging/Resample.c: `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`): separable two-pass convolution with an antialiasing support of 2 x max(scale, 1), double-precision
coefficients normalised per output pixel, converted to 22-bit fixed point, integer accumulation with a rounding half, arithmetic
shift, clamp to [0, 255]; the horizontal pass result is rounded to uint8 before the vertical pass.  A pass whose input and output
length agree is skipped.  Pinned against Pillow itself: tests/golden/resize_bicubic.pt (tests/golden/make_golden_resize.py) and, when
PIL is importable, directly in tests/test_resize.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: coefficients as 22-bit fixed point


def bicubic_filter(x):
    """Keys cubic, a = -0.5 (Resample.c `bicubic_filter`)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support_base=2.0, filt=bicubic_filter):
    """-> (ksize, bounds[out_size, 2] = (first tap, tap count), kk[out_size, ksize] int32)  (Resample.c precompute_coeffs +
    normalize_coeffs_8bpc, box = the full image)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support_base * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8-bit resample pass along `axis` of an [H, W, C] uint8 array."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        acc += np.tensordot(kk[o, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)  # arithmetic shift, then clip8
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_h, out_w):
    """[H, W, C] uint8 -> [out_h, out_w, C] uint8, equal to PIL `Image.resize((out_w, out_h), BICUBIC)` byte for byte."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    if w != out_w:
        _, bx, kx = precompute_coeffs(w, out_w)
        img = _pass(img, bx, kx, axis=1)
    if h != out_h:
        _, by, ky = precompute_coeffs(h, out_h)
        img = _pass(img, by, ky, axis=0)
    return img


def square_transform(img, size=224):
    """square_transform(size)(pil_image) as arrays: [H, W, 3] uint8 -> float32 [3, size, size] in [0, 1] (ToTensor: / 255)."""
    out = resize_bicubic_u8(img, size, size)
    return (out.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
