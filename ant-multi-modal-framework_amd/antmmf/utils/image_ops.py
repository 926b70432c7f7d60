"""Frame transforms named by the *_vtp ymls (`custom_transforms` -> `ImageLongsideScaleAndPad`, `GroupNormalize`), SURVEY.md 8(f4).
Reference: antmmf/utils/image_ops.py:72-108 (GroupNormalize), :127-223 (ImageLongsideScaleAndPad).  Same constructor arguments, same
`random.choice` draw for `random_scale`, same output sizes (`int()` truncation of the short side).  The arithmetic runs on the device
for a whole video at once (antmmf/hip/image.py::frames_bilinear_norm -> csrc/frames.hip); `CustomTransforms` fuses the
scale -> normalise pair of the ymls into ONE pass that can write straight into the padded batch canvas."""
import random

import torch

from antmmf.hip import image as hip_image


def _is_u8_frames(x):
    return isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.dim() == 4


class ImageLongsideScaleAndPad:
    def __init__(self, max_size, random_scale=False, pad=False, interpolation="bilinear", antialias=False):
        """antialias (this build's key): False = torchvision < 0.17's tensor resize (what the reference was written against), True = the tensor default of
        torchvision >= 0.17 (interpolate(..., antialias=True)); both are device kernels (csrc/frames.hip)."""
        assert isinstance(max_size, int)
        self.antialias = bool(antialias)
        if interpolation not in ("bilinear", 2):   # PIL.Image.BILINEAR == 2
            raise NotImplementedError("ImageLongsideScaleAndPad: bilinear only (the default and what every shipped yml uses)")
        if random_scale is False:
            self.scales = [max_size]
        else:
            self.scales = [32 * i for i in range(7, 25) if 32 * i <= max_size]
            if max_size not in self.scales:
                self.scales.append(max_size)
        self.random_scale, self.pad, self.max_size = random_scale, pad, max_size

    def pick_size(self):
        return random.choice(self.scales) if self.random_scale else self.scales[-1]

    @staticmethod
    def get_resize_size(image, max_size):
        """(height, width) of the long side scaled to max_size; tensors are [..., h, w]."""
        height, width = image.shape[-2:]
        if height >= width:
            return int(max_size), int(max_size * (width * 1.0 / height))
        return int(max_size * (height * 1.0 / width)), int(max_size)

    def __call__(self, img, **kwargs):
        """img: frames [n, C, h, w], uint8 (device or emulator) -> float32 resized frames, zero-padded to (max_size, max_size) on the right /
        bottom when pad=True (the reference pads with `self.max_size`, which it never sets: here the scale that was drawn)."""
        if not _is_u8_frames(img):
            raise TypeError("ImageLongsideScaleAndPad: expects the decoder's uint8 frames [n, C, h, w] (float input means the fused uint8 path was bypassed)")
        max_size = self.pick_size()
        oh, ow = self.get_resize_size(img, max_size)
        if not self.pad:
            return hip_image.frames_bilinear_norm(img, oh, ow, antialias=self.antialias)
        canvas = torch.zeros(img.shape[0], img.shape[1], max_size, max_size, dtype=torch.float32, device=img.device)
        hip_image.frames_bilinear_norm(img, oh, ow, out=canvas[:, :, :oh, :ow], antialias=self.antialias)
        return canvas


class GroupNormalize:
    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def channel_stats(self, num_channels):
        if num_channels != len(self.mean):   # e.g. stacked frames: the per-RGB statistics repeat
            return self.mean * (num_channels // len(self.mean)), self.std * (num_channels // len(self.std))
        return self.mean, self.std

    def __call__(self, tensor, **kwargs):
        """In place on a float tensor [bsz, c, h, w] / [c, h, w] (host-side utility; the fused device path is CustomTransforms)."""
        squeeze = tensor.ndim == 3
        if squeeze:
            tensor = tensor.unsqueeze(0)
        c = tensor.size(1)
        mean, std = self.channel_stats(c)
        mean_t = torch.tensor(mean, device=tensor.device).view(1, c, 1, 1)
        std_t = torch.tensor(std, device=tensor.device).view(1, c, 1, 1)
        if torch.max(tensor) > 1 and mean_t.max() <= 1:
            tensor.div_(255.0)
        tensor.sub_(mean_t).div_(std_t)
        return tensor.squeeze(0) if squeeze else tensor
