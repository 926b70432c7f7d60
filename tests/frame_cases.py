"""Shared cases for the video-frame transform (CPU lane emulator and MI355X): product path vs oracle/frames.py (float32: 1e-6 relative)."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import frames as oframes  # noqa: E402  (tests only)

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]   # the *_vtp ymls (detr statistics)


def _close(got, want, what):
    got, want = got.cpu(), want.cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    tol = 1e-6 * max(1.0, want.abs().max().item()) * 4
    assert err <= tol, (what, err, tol)
    return err


def _frames(n, h, w, seed, c=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, c, h, w), dtype=torch.uint8, generator=g)


def case_processor(dev, shapes=((2, 48, 64, 96), (3, 64, 48, 96), (1, 33, 57, 64), (2, 50, 50, 40), (1, 7, 5, 32), (2, 30, 40, 30))):
    """`custom_transforms` [ImageLongsideScaleAndPad, GroupNormalize] as configured by the ymls, up- and down-scaling, odd sizes."""
    import antmmf.datasets.processors  # noqa: F401
    from antmmf.datasets.processors import Processor

    worst = 0.0
    for i, (n, h, w, max_size) in enumerate(shapes):
        cfg = {"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
            {"type": "ImageLongsideScaleAndPad", "params": {"max_size": max_size, "random_scale": False, "pad": False}},
            {"type": "GroupNormalize", "params": {"mean": MEAN, "std": STD}}]}}
        proc = Processor(cfg)
        fr = _frames(n, h, w, 100 + i)
        got = proc(fr.to(dev))
        want = oframes.frame_processor(fr, max_size, MEAN, STD)
        worst = max(worst, _close(got, want, (n, h, w, max_size)))
        assert proc({"image": fr.to(dev)})["image"].shape == want.shape
    # the two transforms alone: resize only (float frames out), and NHWC strides (the decoder's layout) through the same entry point
    from antmmf.hip.image import frames_bilinear_norm
    from antmmf.utils.image_ops import ImageLongsideScaleAndPad

    fr = _frames(2, 40, 56, 7)
    worst = max(worst, _close(ImageLongsideScaleAndPad(80)(fr.to(dev)), oframes.scale_frames(fr, 80), "scale only"))
    nhwc = fr.permute(0, 2, 3, 1).contiguous().to(dev)
    oh, ow = oframes.resize_size(40, 56, 80)
    worst = max(worst, _close(frames_bilinear_norm(nhwc, oh, ow, mean=MEAN, std=STD, layout="nhwc"), oframes.frame_processor(fr, 80, MEAN, STD), "nhwc"))
    padded = ImageLongsideScaleAndPad(80, pad=True)(fr.to(dev)).cpu()
    assert padded.shape == (2, 3, 80, 80) and float(padded[:, :, oh:, :].abs().max()) == 0.0
    return f"max abs err {worst:.2e}"


def case_dark_clip_keeps_reference_semantics(dev):
    """GroupNormalize divides by 255 only when the resized maximum exceeds 1 (image_ops.py:103-104): an all-dark clip (values 0 / 1) is NOT rescaled."""
    import antmmf.datasets.processors  # noqa: F401
    from antmmf.datasets.processors import Processor

    proc = Processor({"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
        {"type": "ImageLongsideScaleAndPad", "params": {"max_size": 32}}, {"type": "GroupNormalize", "params": {"mean": MEAN, "std": STD}}]}})
    dark = (_frames(2, 20, 24, 3) % 2).to(torch.uint8)
    _close(proc(dark.to(dev)), oframes.frame_processor(dark, 32, MEAN, STD), "dark")
    bright = dark.clone(); bright[0, 0, 3, 3] = 200
    _close(proc(bright.to(dev)), oframes.frame_processor(bright, 32, MEAN, STD), "bright")
    # detectron2-style statistics (means on the 0..255 scale): NO division by 255 although the frames exceed 1 (image_ops.py:99-104)
    m255, s255 = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    proc255 = Processor({"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
        {"type": "ImageLongsideScaleAndPad", "params": {"max_size": 32}}, {"type": "GroupNormalize", "params": {"mean": m255, "std": s255}}]}})
    fr = _frames(2, 20, 24, 9)
    _close(proc255(fr.to(dev)), oframes.frame_processor(fr, 32, m255, s255), "means > 1")
    # frames that are not on the library's device / already float take the reference's own host arithmetic (a dataloader worker)
    if dev.type == "cuda":
        _close(proc(bright), oframes.frame_processor(bright, 32, MEAN, STD), "host frames")
    _close(proc(bright.float().to(dev)), oframes.frame_processor(bright, 32, MEAN, STD), "float frames")
    return "ok"


def case_antialias(dev, shapes=((2, 48, 64, 24), (2, 64, 48, 40), (1, 33, 57, 64), (1, 97, 131, 32), (2, 20, 30, 48), (1, 7, 5, 3), (1, 360, 640, 224))):
    """`antialias: true` on ImageLongsideScaleAndPad (torchvision >= 0.17's tensor default) = interpolate(..., antialias=True): down-scaling by integer and
    fractional factors (wide triangle filters), up-scaling (the filter degenerates to plain bilinear taps), tiny outputs, NHWC strides, padded canvas."""
    import antmmf.datasets.processors  # noqa: F401
    from antmmf.datasets.processors import Processor
    from antmmf.hip.image import frames_bilinear_norm
    from antmmf.utils.image_ops import ImageLongsideScaleAndPad

    worst = 0.0
    for i, (n, h, w, max_size) in enumerate(shapes):
        proc = Processor({"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
            {"type": "ImageLongsideScaleAndPad", "params": {"max_size": max_size, "antialias": True}},
            {"type": "GroupNormalize", "params": {"mean": MEAN, "std": STD}}]}})
        fr = _frames(n, h, w, 300 + i)
        worst = max(worst, _close(proc(fr.to(dev)), oframes.frame_processor(fr, max_size, MEAN, STD, antialias=True), ("aa", n, h, w, max_size)))
    fr = _frames(2, 40, 56, 17)
    worst = max(worst, _close(ImageLongsideScaleAndPad(24, antialias=True)(fr.to(dev)), oframes.scale_frames(fr, 24, antialias=True), "aa scale only"))
    oh, ow = oframes.resize_size(40, 56, 24)
    nhwc = fr.permute(0, 2, 3, 1).contiguous().to(dev)
    worst = max(worst, _close(frames_bilinear_norm(nhwc, oh, ow, mean=MEAN, std=STD, layout="nhwc", antialias=True),
                              oframes.frame_processor(fr, 24, MEAN, STD, antialias=True), "aa nhwc"))
    padded = ImageLongsideScaleAndPad(24, pad=True, antialias=True)(fr.to(dev)).cpu()
    assert padded.shape == (2, 3, 24, 24) and float(padded[:, :, oh:, :].abs().max()) == 0.0
    # the filter really is a different one when down-scaling (guards against the flag being dropped on the way)
    plain = ImageLongsideScaleAndPad(24)(fr.to(dev)).cpu()
    assert float((plain - oframes.scale_frames(fr, 24, antialias=True)).abs().max()) > 1.0
    dark = (_frames(2, 20, 24, 3) % 2).to(torch.uint8)    # the device-side `max > 1` test sees the ANTIALIASED values
    procd = Processor({"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
        {"type": "ImageLongsideScaleAndPad", "params": {"max_size": 12, "antialias": True}}, {"type": "GroupNormalize", "params": {"mean": MEAN, "std": STD}}]}})
    _close(procd(dark.to(dev)), oframes.frame_processor(dark, 12, MEAN, STD, antialias=True), "aa dark")
    return f"max abs err {worst:.2e}"


def case_collate(dev, n_clips=2, num_frm=2):
    """A ragged batch of videos -> padded image_data + image_pad_mask, random_scale drawn in the reference's order."""
    import antmmf.datasets.processors  # noqa: F401
    from antmmf.datasets.processors import Processor
    from antmmf.datasets.processors.image_processors import collate_video_frames

    cfg = {"type": "custom_transforms", "params": {"mode": "sequential", "transforms": [
        {"type": "ImageLongsideScaleAndPad", "params": {"max_size": 288, "random_scale": True, "pad": False}},
        {"type": "GroupNormalize", "params": {"mean": MEAN, "std": STD}}]}}
    proc = Processor(cfg)
    vids = [_frames(n_clips * num_frm, h, w, 40 + i) for i, (h, w) in enumerate(((36, 64), (64, 36), (48, 48)))]
    random.seed(21)
    data, mask = collate_video_frames([v.to(dev) for v in vids], proc.processor, n_clips=n_clips, num_frm=num_frm)
    random.seed(21)
    want_d, want_m = oframes.collate([oframes.frame_processor(v, 288, MEAN, STD, random_scale=True) for v in vids])
    _close(data, want_d, "collate data")
    assert torch.equal(mask.cpu(), want_m)
    return tuple(data.shape)


def case_full_size(dev, n=12, h=360, w=640, max_size=448):
    """One 12-frame 640x360 video at the test-time size of the ymls (max_size 448): oracle on two frames, properties on all."""
    from antmmf.hip.image import frames_bilinear_norm

    fr = _frames(n, h, w, 5)
    oh, ow = oframes.resize_size(h, w, max_size)
    got = frames_bilinear_norm(fr.to(dev), oh, ow, mean=MEAN, std=STD)
    _close(got[:2], oframes.frame_processor(fr[:2], max_size, MEAN, STD), "full size")
    flat = torch.full((1, 3, h, w), 128, dtype=torch.uint8)
    const = frames_bilinear_norm(flat.to(dev), oh, ow, mean=MEAN, std=STD).cpu()   # a constant frame stays constant
    for c in range(3):
        assert float((const[0, c] - (128 / 255 - MEAN[c]) / STD[c]).abs().max()) < 1e-5
    return (oh, ow)
