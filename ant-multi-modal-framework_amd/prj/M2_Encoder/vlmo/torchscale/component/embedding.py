"""Vision / text / positional embedding parameter holders (reference:
prj/M2_Encoder/vlmo/torchscale/component/embedding.py:30-110); the arithmetic is antmmf.hip.functional.patch_embed /
embed, called from BEiT3 / Encoder."""
import torch
from torch import nn


class VisionEmbedding(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, contain_mask_token=False, prepend_cls_token=False):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)  # parameter holder
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if contain_mask_token else None
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if prepend_cls_token else None

    def num_position_embeddings(self):
        return self.num_patches if self.cls_token is None else self.num_patches + 1


class TextEmbedding(nn.Embedding):
    def reset_parameters(self):
        nn.init.normal_(self.weight, mean=0, std=self.embedding_dim ** -0.5)
        self._fill_padding_idx_with_zero()


class PositionalEmbedding(nn.Embedding):
    """positions start at 2 (Fairseq convention, reference :92-110)."""
