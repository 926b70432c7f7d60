"""`custom_transforms` (reference antmmf/datasets/processors/image_processors.py:447-547): a list of transforms from
antmmf.utils.image_ops applied in sequence (or one at random).  For the frame processors of the *_vtp ymls --
[ImageLongsideScaleAndPad, GroupNormalize] over a video's uint8 frames -- the pair runs as ONE device pass
(`fused_scale_normalize`): no float copy of the input, no intermediate resized tensor, optional direct write into the padded batch."""
import random

import torch

from antmmf.common.registry import registry
from antmmf.hip import image as hip_image
from antmmf.utils import image_ops

from .processors import BaseProcessor, _get


@registry.register_processor("custom_transforms")
class CustomTransforms(BaseProcessor):
    def __init__(self, config, *args, **kwargs):
        self.config = config
        self.mode = _get(config, "mode")
        assert self.mode in ["sequential", "random"]
        transforms = _get(config, "transforms")
        assert isinstance(transforms, (list, tuple))
        self.transfunc_list, self.transfunc_params = [], []
        for t in transforms:
            ttype, tparams = t["type"], (t.get("params", {}) or {})
            obj = getattr(image_ops, ttype, None)
            assert obj is not None, f"antmmf.utils.image_ops has no transform: {ttype}"
            self.transfunc_list.append(obj(**tparams) if isinstance(obj, type) else obj)
            self.transfunc_params.append(dict(tparams))

    def _fusable(self):
        f = self.transfunc_list
        return (self.mode == "sequential" and len(f) == 2 and isinstance(f[0], image_ops.ImageLongsideScaleAndPad)
                and isinstance(f[1], image_ops.GroupNormalize))

    @staticmethod
    def _on_library_device(x):
        from antmmf.hip import _lib

        try:
            return (x.device.type == "cuda") == (_lib.backend() == 1)
        except Exception:   # no library at all (a host-only data worker)
            return False

    def output_size(self, x):
        """(out_h, out_w) the scale transform gives frames `x` -- draws the random scale exactly once, like one reference call."""
        scale = self.transfunc_list[0]
        return scale.get_resize_size(x, scale.pick_size())

    def fused_scale_normalize(self, x, size=None, out=None):
        scale, norm = self.transfunc_list
        oh, ow = size if size is not None else self.output_size(x)
        mean, std = norm.channel_stats(x.shape[1])
        if scale.pad:
            raise NotImplementedError("custom_transforms: pad=True inside the fused path (every shipped yml has pad: false)")
        # GroupNormalize divides by 255 only when the frames exceed 1 AND the means are on the [0, 1] scale (image_ops.py:99-104: detectron2-style
        # means such as 123.675 mean "pixels stay on 0..255"): the second half is a host-side property of the configuration, the first stays on the device
        div255 = -1 if max(mean) <= 1 else 0
        return hip_image.frames_bilinear_norm(x, oh, ow, mean=mean, std=std, out=out, div255=div255, antialias=scale.antialias)

    def __call__(self, x):
        return_dict = isinstance(x, dict)
        if return_dict:
            x = x["image"]
        idx = None
        if self._fusable() and isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.dim() == 4 and self._on_library_device(x):
            res = self.fused_scale_normalize(x)
        elif self._fusable() and isinstance(x, torch.Tensor) and x.dim() == 4:
            # frames that are not on the library's device (a dataloader worker of the reference transforms on the host) or already float: the reference's own
            # arithmetic -- torchvision's tensor resize IS torch.nn.functional.interpolate(bilinear, align_corners=False) -- then GroupNormalize
            scale, norm = self.transfunc_list
            if scale.pad:
                raise NotImplementedError("custom_transforms: pad=True (every shipped yml has pad: false)")
            oh, ow = self.output_size(x)
            res = norm(torch.nn.functional.interpolate(x.float(), size=(oh, ow), mode="bilinear", align_corners=False, antialias=scale.antialias))
        elif self.mode == "sequential":
            res = x
            for func, param in zip(self.transfunc_list, self.transfunc_params):
                res = func(res, **param)
        else:
            n = len(self.transfunc_list)
            idx = _get(self.config, "idx", None)
            if idx is None:
                idx = random.randint(0, n)
            res = self.transfunc_list[idx](x, **self.transfunc_params[idx]) if idx < n else x
        if return_dict:
            res = {"image": res}
            if idx is not None:
                res["idx"] = idx
        return res


def collate_video_frames(videos, processor, n_clips, num_frm, device=None):
    """The frame half of MMFUnivlVideoDataset.get_item + collate_fn (prj/base_vtp/roi_univl/univl/video_text/ret_dataset.py:97-115,176-199) for a whole
    batch: `videos` = one uint8 [n_clips * num_frm, C, h, w] tensor per sample (any sizes).  Every video is transformed by ONE launch pair that
    writes into its corner of the zero-initialised canvas -> (image_data float32 [B, n_clips * num_frm, C, H, W], image_pad_mask bool
    [B, n_clips * num_frm, H, W], True = padding) with H / W the batch maxima, exactly what NestedTensor.from_tensor_list + view gives."""
    if not videos:
        raise ValueError("collate_video_frames: empty batch")
    if device is not None:
        videos = [v.to(device, non_blocking=True) for v in videos]
    nf = n_clips * num_frm
    for v in videos:
        if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[0] != nf:
            raise ValueError(f"collate_video_frames: every video must be uint8 [{nf}, C, h, w]")
    sizes = [processor.output_size(v) for v in videos]   # one random-scale draw per video, in batch order
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    C, dev = videos[0].shape[1], videos[0].device
    data = torch.zeros(len(videos), nf, C, H, W, dtype=torch.float32, device=dev)
    mask = torch.ones(len(videos), nf, H, W, dtype=torch.bool, device=dev)
    for b, (v, (oh, ow)) in enumerate(zip(videos, sizes)):
        processor.fused_scale_normalize(v, size=(oh, ow), out=data[b, :, :, :oh, :ow])
        mask[b, :, :oh, :ow] = False
    return data, mask
