"""square_transform (reference: prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14): Resize((size, size), BICUBIC) + ToTensor.

Same name and call convention -- `square_transform(size)(pil_image) -> float32 [3, size, size]` in [0, 1] -- but the resize runs on
the MI355X (antmmf.hip.image.resize_bicubic_u8, byte-identical to Pillow's) and the result stays in HBM, ready for
`VLMo.infer_image`, which applies the inception normalisation inside the patch-extraction kernel.  `.batch(images)` resizes a whole
ragged batch with two kernel launches instead of one PIL call per image.  The random-augmentation variant is a training-time CPU
augmentation outside the hot path.
"""
import numpy as np
import torch

from antmmf.hip.image import resize_bicubic_u8


def _as_u8_hwc(img):
    if isinstance(img, torch.Tensor):
        t = img
    else:  # PIL.Image or anything numpy can view as [h, w, 3] uint8 (no PIL import needed here)
        arr = np.asarray(img)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr))
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise TypeError("square_transform expects an 8-bit [h, w, c] image (PIL.Image, ndarray or tensor)")
    return t


class SquareTransform:
    def __init__(self, size=224):
        self.size = int(size)

    def __call__(self, img):
        return self.batch([img])[0]

    def batch(self, images):
        return resize_bicubic_u8([_as_u8_hwc(i) for i in images], self.size, self.size, out_f32=True)

    def __repr__(self):
        return f"SquareTransform(size={self.size}, interpolation=bicubic, device=hip)"


def square_transform(size=224):
    return SquareTransform(size)


def square_transform_randaug(size=224):
    raise NotImplementedError("square_transform_randaug is a CPU training augmentation (RandomResizedCrop + RandAugment): out of scope")
