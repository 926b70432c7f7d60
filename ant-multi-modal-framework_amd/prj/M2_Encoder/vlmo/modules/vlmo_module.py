"""VLMo: the M2-Encoder dual tower (BEiT-3 multiway) on the MI355X path.

Reference: prj/M2_Encoder/vlmo/modules/vlmo_module.py:130-405 -- an inference-only release: `infer_image` /
`infer_text` and the ITC heads + logit scales exist there, a training step does not (SURVEY.md "Facts" 2).
This class keeps the constructor keys, sub-module / parameter names (state_dicts load unchanged) and the
infer_* contracts, and adds the training step the metric is defined on:
    forward(batch) -> {"losses": {"itc_loss", "itc_vl_loss"}, "logits"...}
with symmetric InfoNCE over logit_scale.exp() * img @ txt.T (the reference's logits formula,
prj/M2_Encoder/m2_encoder.py:92-95) on both the `cls_feats` and `cls_vlffn_feats` pairs, global negatives gathered
over RCCL and the loss row-sharded (antmmf.hip.contrastive).  Tokeniser, transforms, checkpoint conversion and the
Lightning plumbing are outside the step path.
"""
import math

import os

import numpy as np
import torch
from torch import nn

from antmmf.hip import contrastive
from antmmf.hip import functional as HF
from . import heads
from .modeling_utils import BEiT3, get_config
from ..torchscale.architecture.encoder import Encoder


def _resize_pos_embed(value, n_special, num_visual_token):
    """Area-interpolate the square patch grid of a position table to sqrt(num_visual_token - 1)^2 cells; the first `n_special`
    rows (cls / special positions) are kept."""
    n_old = value.shape[0] - n_special
    dim = value.shape[-1]
    side_old, side_new = int(math.sqrt(n_old)), int(math.sqrt(num_visual_token - 1))
    special, patch = value[:n_special], value[n_special:].float()
    patch = nn.functional.interpolate(patch.reshape(1, side_old, side_old, dim).permute(0, 3, 1, 2), size=(side_new, side_new), mode="area")
    patch = patch.to(special.dtype).permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat((special, patch), dim=0)


def convert_pl_ckpt(state_dict, num_visual_token=197):
    """Lightning checkpoint -> this model (reference vlmo_module.py:22-56): drop the visual tokenizer, and bring
    backbone.encoder.embed_positions.A.weight (3 special rows + patch grid) to `num_visual_token + 2` rows -- area interpolation of
    the grid when the checkpoint has fewer rows, truncation when it has more."""
    new_state_dict = {}
    for key, value in state_dict.items():
        if "visual_tokenizer" in key:
            continue
        if "backbone.encoder.embed_positions.A.weight" in key:
            if value.shape[0] < num_visual_token + 2:
                value = _resize_pos_embed(value, 3, num_visual_token)
            elif value.shape[0] > num_visual_token + 2:
                value = value[:num_visual_token + 2, :]
        new_state_dict[key] = value
    return new_state_dict


def convert_deepspeed_ckpt(state_dict, num_visual_token=197):
    """DeepSpeed checkpoint -> this model (reference :59-106): strip the `_forward_module.` prefix; resize the visual tokenizer's
    [1, 1 + grid, dim] position tables and the backbone's position table (same rule as convert_pl_ckpt, interpolation only)."""
    new_state_dict = {}
    for key, value in state_dict.items():
        if not key.startswith("_forward_module."):
            new_state_dict[key] = value
            continue
        new_key = key[len("_forward_module."):]
        if ("visual_tokenizer.encoder.pos_embed" in new_key or "visual_tokenizer.decoder.pos_embed" in new_key) and value.shape[1] != num_visual_token:
            value = _resize_pos_embed(value[0], 1, num_visual_token).unsqueeze(0)
        if "backbone.encoder.embed_positions.A.weight" in new_key and value.shape[1] != num_visual_token + 2:
            # (the reference compares shape[1] -- the embedding width -- so this branch runs for every checkpoint; a table that already
            # has the target grid comes back unchanged from the area interpolation)
            value = _resize_pos_embed(value, 3, num_visual_token)
        new_state_dict[new_key] = value
    return new_state_dict


def init_weights(module):
    """objectives.init_weights of the reference: N(0, 0.02) for Linear / Embedding, unit LayerNorm."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


def get_pretrained_tokenizer(tokenizer_type, from_pretrained):
    """reference :121-127 (`eval(tokenizer_type).from_pretrained(path)`), as a table: GLMChineseTokenizer = sp.model directory of the released
    checkpoints; BertTokenizer = a vocab.txt (the config.py default).  No rank-0-first barrier: nothing is downloaded."""
    if tokenizer_type == "GLMChineseTokenizer":
        from vlmo.tokenizer.tokenization_glm import GLMChineseTokenizer

        return GLMChineseTokenizer.from_pretrained(from_pretrained)
    if tokenizer_type == "BertTokenizer":
        from antmmf.datasets.tokenization import BertWordPieceTokenizer

        path = os.path.join(from_pretrained, "vocab.txt") if os.path.isdir(from_pretrained) else from_pretrained
        return BertWordPieceTokenizer(path, do_lower_case="uncased" in os.path.basename(os.path.normpath(from_pretrained)))
    raise NotImplementedError(f"get_pretrained_tokenizer: {tokenizer_type!r} (GLMChineseTokenizer | BertTokenizer)")


class VLMo(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.hparams = type("HP", (), {"config": config})()
        self.img_size = config["image_size"]
        kwargs = {}
        if "encoder_attention_heads" in config:
            kwargs["encoder_attention_heads"] = config["encoder_attention_heads"]
        args = get_config(
            config["beit_version"], img_size=config["image_size"], patch_size=config["patch_size"], vocab_size=config["vocab_size"],
            encoder_layers=config["encoder_layers"], encoder_embed_dim=config["encoder_embed_dim"],
            checkpoint_activations=config.get("checkpoint_activations", False), share_layer=config.get("share_layer", False),
            share_attn=config.get("share_attn", False), deepnorm=config.get("deepnorm", False), mask_ratio=config.get("mask_ratio", 0),
            max_text_len=config.get("max_text_len", 52), one_attn=config.get("one_attn", False), **kwargs)
        self.num_features = args.encoder_embed_dim
        self.out_features = config["out_embed_dim"]
        self.patch_size = config["patch_size"]
        self.num_frames = config.get("num_frames", 1)
        # host-side tokenizer (reference :167-168): built only when the config names one -- the training step takes ids
        self.tokenizer_type = config.get("tokenizer_type", None)
        self.text_tokenizer = get_pretrained_tokenizer(self.tokenizer_type, config["tokenizer"]) if (self.tokenizer_type and config.get("tokenizer")) else None
        self.backbone = BEiT3(args)
        self.use_vl = config["beit3_vl_layers"] > 0
        if self.use_vl:
            args.encoder_layers = config["beit3_vl_layers"]
            self.backbone_vl = Encoder(args)
        self.norm = nn.LayerNorm(self.num_features, eps=1e-6)
        self.pooler = heads.Pooler(self.num_features)
        self.pooler.apply(init_weights)
        if config["loss_names"]["itc"] > 0:
            for name in ("itc_text_proj", "itc_image_proj", "itc_vl_text_proj", "itc_vl_image_proj"):
                head = heads.ITCHead(self.num_features, self.out_features)
                head.apply(init_weights)
                setattr(self, name, head)
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
            self.logit_vl_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.backbone.apply(init_weights)
        if self.use_vl:
            self.backbone_vl.apply(init_weights)
        self._local_loss = config.get("local_loss", False)
        self._aggregate_nodes = config.get("aggregate_nodes", -1)
        if config.get("load_path", "") != "" and config.get("test_only", False):
            self.load_checkpoint(config["load_path"])

    def load_checkpoint(self, path):
        """Released-weight loading (reference :207-232): Lightning ("state_dict"), DeepSpeed ("module" / `_forward_module.` keys) or
        plain state dicts, position tables resized to this model's patch grid, strict=False."""
        ckpt = torch.load(path, map_location="cpu")
        n_pos = self.backbone.vision_embed.num_position_embeddings()
        state_dict = None
        for k in ("state_dict", "module", "model"):
            if k in ckpt:
                state_dict = ckpt[k]
                if k == "module":
                    state_dict = convert_deepspeed_ckpt(state_dict, n_pos)
                elif k == "state_dict":
                    state_dict = convert_pl_ckpt(state_dict, n_pos)
                break
        if state_dict is None:
            state_dict = convert_deepspeed_ckpt(ckpt, n_pos) if next(iter(ckpt)).startswith("_forward_module.") else ckpt
        return self.load_state_dict(state_dict, strict=False)

    # ------------------------------------------------------------------ towers (reference :323-405)
    def infer_text(self, batch, mask_text=False):
        assert not mask_text, "MLM is outside the ITC path"
        text_ids, text_masks = batch["text_ids"], batch["text_masks"]
        pad = 1 - text_masks
        lffn = self.backbone(textual_tokens=text_ids, text_padding_position=pad)["encoder_out"]
        vlffn = self.backbone_vl(token_embeddings=lffn, encoder_padding_mask=pad.bool(), multiway_split_position=-1)["encoder_out"]
        cls_feats = HF.l2_normalize(self.itc_text_proj(lffn[:, 0]), eps=0.0)
        cls_vlffn_feats = HF.l2_normalize(self.itc_vl_text_proj(vlffn[:, 0]), eps=0.0)
        return {"cls_feats": cls_feats, "cls_vlffn_feats": cls_vlffn_feats, "text_feats": lffn}

    def infer_image(self, batch, mask_image=False, image_token_type_idx=1, image_embeds=None, image_masks=None):
        assert not mask_image, "image MLM is outside the ITC path"
        key = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
        img = batch[key][0]
        # inception normalise (x - 0.5) / 0.5 is fused into the patch extraction (reference: img_norm, :385)
        vffn = self.backbone(visual_tokens=img, image_shift=0.5, image_scale=2.0)["encoder_out"]
        vlffn = self.backbone_vl(token_embeddings=vffn, multiway_split_position=-1)["encoder_out"]
        cls_feats = HF.l2_normalize(self.itc_image_proj(vffn[:, 0]), eps=0.0)
        cls_vlffn_feats = HF.l2_normalize(self.itc_vl_image_proj(vlffn[:, 0]), eps=0.0)
        return {"image_feats": vffn, "cls_feats": cls_feats, "cls_vlffn_feats": cls_vlffn_feats}

    # ------------------------------------------------------------------ training step (this build's; see module docstring)
    def compute_itc(self, batch):
        oi, ot = self.infer_image(batch), self.infer_text(batch)
        l1, l2 = contrastive.clip_itc_pair_sharded(oi["cls_feats"], ot["cls_feats"], self.logit_scale,
                                                   oi["cls_vlffn_feats"], ot["cls_vlffn_feats"], self.logit_vl_scale)
        return {"losses": {"itc_loss": 0.5 * l1, "itc_vl_loss": 0.5 * l2}, "image": oi, "text": ot}

    def forward(self, batch):
        return self.compute_itc(batch)
