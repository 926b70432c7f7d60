"""Input-pipeline step in front of the image tower: batched, Pillow-exact bicubic resize on the device.

Reference: `square_transform(size)` = torchvision Resize((size, size), BICUBIC) + ToTensor, applied per image on the CPU
(prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14; prj/M2_Encoder/m2_encoder.py:61-68); the arithmetic is Pillow's
src/libImaging/Resample.c.  Here the host only builds the coefficient tables (a few KB per distinct image extent, cached) and one
descriptor row per image; the byte work for the whole ragged batch is two kernel launches (csrc/resize.hip).
"""
import functools
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _rc, _stream

PRECISION_BITS = 22  # Resample.c: 32 - 8 - 2


@functools.lru_cache(maxsize=4096)
def bicubic_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (a = -0.5, support 2) over the full extent, in
    the same double-precision statement order (the per-pixel weight sum is accumulated tap by tap, not pairwise).
    -> (taps per output, bounds int32 [out, 2] = (first tap, tap count), coeffs int32 [out, taps])"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    a = -0.5
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.abs((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
        ww = float(np.cumsum(w)[-1]) if xmax > 0 else 0.0  # sequential sum, like the C loop
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = np.trunc(np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    if int(np.abs(kk).max()) >= 1 << 23:  # the kernels multiply with v_mad_i32_i24
        raise ValueError(f"bicubic_coeffs({in_size}, {out_size}): coefficient outside the 24-bit range")
    return ksize, bounds, kk


class ResizePlan:
    """Descriptor + coefficient tables of one ragged batch, on the device.  Building it costs a few host microseconds per image and
    three small uploads; a dataloader that sees the same frame sizes again can keep it."""

    def __init__(self, sizes, channels, out_h, out_w, device):
        desc = np.zeros((len(sizes), 10), dtype=np.int64)
        tabs, btabs, where = [], [], {}
        n_coef = n_bound = 0

        def table(in_size, out_size, tap_major):
            nonlocal n_coef, n_bound
            if in_size == out_size:  # Pillow skips a pass whose extent does not change
                return 0, 0, 0
            key = (in_size, out_size, tap_major)
            if key not in where:
                ks, b, k = bicubic_coeffs(in_size, out_size)
                where[key] = (n_coef, ks, n_bound)
                if tap_major:  # [taps rounded up to 4, out] with zero rows: the RGB kernel consumes four taps per iteration
                    kt = np.zeros(((ks + 3) // 4 * 4, k.shape[0]), dtype=np.int32)
                    kt[:ks] = k.T
                    k = kt
                tabs.append(k.reshape(-1)); btabs.append(b.reshape(-1))
                n_coef += k.size; n_bound += b.size
            return where[key]

        src_off = tmp_off = 0
        for i, (h, w) in enumerate(sizes):
            kx_off, kx, bx_off = table(w, out_w, True)    # horizontal pass: lanes = outputs -> tap-major table
            ky_off, ky, by_off = table(h, out_h, False)   # vertical pass: a wave shares one output row
            desc[i] = (src_off, h, w, tmp_off, kx_off, kx, bx_off, ky_off, ky, by_off)
            src_off += h * w * channels
            tmp_off += h * out_w * channels
        self.n, self.channels, self.out_h, self.out_w = len(sizes), channels, out_h, out_w
        self.src_bytes, self.tmp_bytes = src_off, tmp_off
        self.max_h, self.max_w = int(desc[:, 1].max()), int(desc[:, 2].max())
        self.coeffs = torch.from_numpy(np.concatenate(tabs) if tabs else np.zeros(1, np.int32)).to(device)
        self.bounds = torch.from_numpy(np.concatenate(btabs) if btabs else np.zeros(2, np.int32)).to(device)
        self.desc = torch.from_numpy(desc).to(device)
        self.device = torch.device(device)


def resize_packed_u8(src, plan, out_f32=True, tmp=None):
    """src: uint8 device buffer holding the plan's images back to back ([h_i, w_i, C]); two kernel launches."""
    if src.dtype != torch.uint8 or not src.is_contiguous() or src.numel() != plan.src_bytes or src.device != plan.device:
        raise ValueError("resize_packed_u8: src must be the contiguous uint8 buffer the plan describes, on the plan's device")
    if tmp is None or tmp.numel() < plan.tmp_bytes:
        tmp = torch.empty(max(plan.tmp_bytes, 4), dtype=torch.uint8, device=plan.device)
    out = (torch.empty(plan.n, plan.channels, plan.out_h, plan.out_w, dtype=torch.float32, device=plan.device) if out_f32
           else torch.empty(plan.n, plan.out_h, plan.out_w, plan.channels, dtype=torch.uint8, device=plan.device))
    _rc(_lib.load().antmmf_resize_bicubic_u8(_p(src), src.numel(), _p(plan.desc), plan.n, plan.max_h, plan.max_w, plan.channels, plan.out_h,
                                            plan.out_w, _p(plan.coeffs), _p(plan.bounds), _p(tmp), _p(out), 1 if out_f32 else 0, _stream()),
        "antmmf_resize_bicubic_u8")
    return out


def resize_bicubic_u8(images, out_h, out_w, out_f32=True, device=None):
    """images: sequence of uint8 [h, w, C] tensors (any sizes, same C; host tensors are uploaded) ->
    float32 [n, C, out_h, out_w] = resized / 255 (out_f32, what Resize + ToTensor returns) or uint8 [n, out_h, out_w, C]
    (byte-identical to PIL `Image.resize((out_w, out_h), BICUBIC)`)."""
    if len(images) == 0:
        raise ValueError("resize_bicubic_u8: empty batch")
    C = images[0].shape[2]
    for t in images:
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != C or t.shape[0] < 1 or t.shape[1] < 1:
            raise ValueError("resize_bicubic_u8: images must be uint8 [h, w, C] with a common channel count")
    if device is None:
        if _lib.backend() == 1:  # decoded frames arrive in host memory: they are uploaded, the resize runs on the GPU
            device = next((t.device for t in images if t.is_cuda), torch.device("cuda", torch.cuda.current_device()))
        else:
            device = torch.device("cpu")  # CPU lane emulator (tests)
    if (torch.device(device).type == "cuda") != (_lib.backend() == 1):
        raise RuntimeError("resize_bicubic_u8: device does not match the loaded library (gfx950 library <-> cuda tensors; no CPU fallback)")
    plan = ResizePlan([(int(t.shape[0]), int(t.shape[1])) for t in images], C, out_h, out_w, device)
    src = torch.cat([t.reshape(-1).to(device, non_blocking=True) for t in images])
    return resize_packed_u8(src, plan, out_f32)


# ------------------------------------------------------------------------------ video frames (bilinear, float32, fused normalise + pad)
def frames_bilinear_norm(frames, out_h, out_w, mean=None, std=None, out=None, div255=-1, layout="nchw", antialias=False):
    """All frames of one video: uint8 `frames` ([n, C, h, w], or [n, h, w, C] with layout="nhwc": only the strides differ) ->
    float32 [n, C, out_h, out_w] = GroupNormalize(bilinear_resize(frames.float())) (csrc/frames.hip; reference
    image_processors.py:520-547 + image_ops.py:72-108,127-223).  `out`: optional float32 view to write into (last dim contiguous) --
    e.g. the [n, C, :out_h, :out_w] corner of the zero-initialised padded batch canvas, which makes the collate padding free.
    mean / std None: resize only.  div255 -1: the reference's `max > 1` test, evaluated on the device.
    antialias: torchvision >= 0.17's tensor default (interpolate(..., antialias=True): ATen's separable triangle filter, two passes)."""
    if frames.dtype != torch.uint8 or frames.dim() != 4:
        raise ValueError("frames_bilinear_norm: frames must be a uint8 [n, C, h, w] (or [n, h, w, C]) tensor")
    if (frames.device.type == "cuda") != (_lib.backend() == 1):
        raise RuntimeError("frames_bilinear_norm: device does not match the loaded library (gfx950 library <-> cuda tensors; no CPU fallback)")
    if layout == "nhwc":
        n, h, w, C = frames.shape
        sn, sh, sw, sc = frames.stride()
    else:
        n, C, h, w = frames.shape
        sn, sc, sh, sw = frames.stride()
    dev = frames.device
    if out is None:
        out = torch.empty(n, C, out_h, out_w, dtype=torch.float32, device=dev)
    if out.dtype != torch.float32 or tuple(out.shape) != (n, C, out_h, out_w) or out.stride(3) != 1 or out.device != dev:
        raise ValueError("frames_bilinear_norm: out must be a float32 [n, C, out_h, out_w] view with contiguous columns on the frames' device")
    mean_t = std_t = scratch = None
    if mean is not None:
        mean_t = torch.as_tensor(mean, dtype=torch.float32).reshape(-1).to(dev)
        std_t = torch.as_tensor(std, dtype=torch.float32).reshape(-1).to(dev)
        if mean_t.numel() != C or std_t.numel() != C:
            raise ValueError("frames_bilinear_norm: mean / std need one value per channel")
        if div255 < 0:
            scratch = torch.zeros(1, dtype=torch.int32, device=dev)
    if antialias:
        temp = torch.empty(n * C * h * out_w, dtype=torch.float32, device=dev)   # the horizontally filtered frames
        _rc(_lib.load().antmmf_frames_bilinear_aa_norm(_p(frames), n, C, h, w, sn, sc, sh, sw, _p(temp), _p(out), out_h, out_w, out.stride(0), out.stride(1),
                                                      out.stride(2), _p(mean_t), _p(std_t), int(div255), _p(scratch), _stream()), "antmmf_frames_bilinear_aa_norm")
        return out
    _rc(_lib.load().antmmf_frames_bilinear_norm(_p(frames), n, C, h, w, sn, sc, sh, sw, _p(out), out_h, out_w, out.stride(0), out.stride(1), out.stride(2),
                                               _p(mean_t), _p(std_t), int(div255), _p(scratch), _stream()), "antmmf_frames_bilinear_norm")
    return out
