"""VitImageEncoder: CLIP ViT registered into the VisualEncoder family (reference:
prj/base_vtp/roi_univl/univl/model/clip_visual_encoder.py:15-94).  Contract kept: forward(image [B,T,C,H,W],
image_mask [B,T,H,W] bool) -> {grid_feature [B,T,out_dim,1,1], grid_mask [B,T,1,1], grid_feature_with_pos: None},
attribute `out_dim`."""
import os

import torch
from torch import nn

from antmmf.modules.encoders import VisualEncoder
from antmmf.modules.vision.backbone.clip.model import VisionTransformer


@VisualEncoder.register()
class VitImageEncoder(nn.Module):
    def __init__(self, model_name: str, input_resolution: int, patch_size: int, width: int, layers: int, out_dim: int,
                 head_width=64, pretrained=True, is_proj=True):
        super().__init__()
        self.visual = VisionTransformer(input_resolution=input_resolution, patch_size=patch_size, width=width, layers=layers,
                                        heads=width // head_width, output_dim=out_dim)
        if not is_proj:
            self.visual.proj = None
        self.out_dim = out_dim
        if pretrained:
            self.load_pretrained(model_name)

    def load_pretrained(self, name):
        """CN-CLIP checkpoint: keep `visual.*` keys, strip a leading `module.` (reference :46-71)."""
        if not os.path.isfile(name):
            raise RuntimeError(f"Model {name} not found (no network here: pass a local checkpoint path or pretrained=False)")
        sd = torch.load(name, map_location="cpu")["state_dict"]
        picked = {}
        for k, v in sd.items():
            if "visual" not in k:
                continue
            k = k[len("module."):] if k.startswith("module.") else k
            picked[k[len("visual."):] if k.startswith("visual.") else k] = v
        self.visual.load_state_dict(picked, strict=True)

    def forward(self, image, image_mask):
        b, t, c, h, w = image.shape
        feat = self.visual(image.reshape(b * t, c, h, w))
        feat = feat.view(b, t, self.out_dim, 1, 1)
        # the 1x1 "grid" of a ViT frame is padding only if the whole frame is padding (nearest interpolation to 1x1
        # picks pixel (0, 0): reference :89)
        mask = image_mask[:, :, :1, :1].to(torch.bool)
        return dict(grid_feature=feat, grid_mask=mask, grid_feature_with_pos=None)
