"""TEST INFRASTRUCTURE (oracle): CPU restatement of the video-frame transform in front of the visual tower.

Reference path, per video, on the dataloader workers:
  CustomTransforms.__call__ (antmmf/datasets/processors/image_processors.py:520-547): uint8 frames [n, C, h, w] -> float32, then in sequence
  ImageLongsideScaleAndPad (antmmf/utils/image_ops.py:127-223): `torchvision.transforms.functional.resize(frames, (h', w'), BILINEAR)`,
    (h', w') = long side -> max_size, short side int(max_size * short / long); max_size drawn by random.choice when random_scale
  GroupNormalize (antmmf/utils/image_ops.py:72-108): `/ 255` if the tensor's maximum is > 1 (and the means are <= 1), `- mean`, `/ std`
and at collate time NestedTensor.from_tensor_list (antmmf/structures/nested_tensor.py:51-63 -> structures/utils.py:40-120): zero-pad every
frame to the batch maxima, mask True over the padding.

The resize lives in a third-party dependency that is not installed in this image (**torchvision**, requirements.txt `torchvision`
unpinned): for a float tensor its `resize` is `torch.nn.functional.interpolate(x, size, mode="bilinear", align_corners=False)` -- without
antialiasing in the torchvision generations contemporary with the reference (antialias became the tensor default only in 0.17; the
`antialias=True` argument below restates that later default: the same function with antialias=True).  torch
IS installed, so the oracle calls that very function; **the torchvision wrapper itself is unpinned** (stated in DESIGN.md).
"""
import random

import torch
import torch.nn.functional as F


def resize_size(h, w, max_size):
    """image_ops.py:191-223."""
    if h >= w:
        return int(max_size), int(max_size * (w * 1.0 / h))
    return int(max_size * (h * 1.0 / w)), int(max_size)


def scales(max_size, random_scale):
    """image_ops.py:154-159."""
    if random_scale is False:
        return [max_size]
    s = [32 * i for i in range(7, 25) if 32 * i <= max_size]
    if max_size not in s:
        s.append(max_size)
    return s


def scale_frames(frames_u8, max_size, random_scale=False, antialias=False):
    """CustomTransforms' float cast + ImageLongsideScaleAndPad (pad=False) -> float32 [n, C, h', w']."""
    sc = scales(max_size, random_scale)
    m = random.choice(sc) if random_scale else sc[-1]
    x = frames_u8.float()
    return F.interpolate(x, size=resize_size(x.shape[-2], x.shape[-1], m), mode="bilinear", align_corners=False, antialias=antialias)


def group_normalize(x, mean, std):
    """image_ops.py:77-108 on a float [n, C, h, w] tensor (in place on a copy)."""
    x = x.clone()
    c = x.size(1)
    if c != len(mean):
        mean, std = list(mean) * (c // len(mean)), list(std) * (c // len(std))
    m = torch.tensor(mean).view(1, c, 1, 1)
    s = torch.tensor(std).view(1, c, 1, 1)
    if torch.max(x) > 1 and m.max() <= 1:
        x.div_(255.0)
    return x.sub_(m).div_(s)


def frame_processor(frames_u8, max_size, mean, std, random_scale=False, antialias=False):
    return group_normalize(scale_frames(frames_u8, max_size, random_scale, antialias), mean, std)


def collate(videos):
    """ret_dataset.py:176-199: list of [nf, C, h_i, w_i] float tensors -> (data [B, nf, C, H, W], mask [B, nf, H, W] True = padding)."""
    nf, C = videos[0].shape[:2]
    H, W = max(v.shape[-2] for v in videos), max(v.shape[-1] for v in videos)
    data = torch.zeros(len(videos), nf, C, H, W)
    mask = torch.ones(len(videos), nf, H, W, dtype=torch.bool)
    for b, v in enumerate(videos):
        data[b, :, :, :v.shape[-2], :v.shape[-1]] = v
        mask[b, :, :v.shape[-2], :v.shape[-1]] = False
    return data, mask
