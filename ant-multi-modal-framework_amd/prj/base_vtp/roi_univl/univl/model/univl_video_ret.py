"""UnivlForVideoTextRetrieval, stage 1 (ITC / MIL-NCE) on the MI355X path (reference:
prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:16-31,146-226,251-387,445-542).

Training step: towers -> L2-normalised embeddings -> row-sharded global MIL-NCE (antmmf.hip.contrastive), which
replaces gather_tensor x2 + get_l1_simi_matrix + the tiled [T*n, V*n] matrix + get_mil_nce_loss.  `l1_simi`
(reported [T, V] scores, logsumexp over clips) is produced for the LOCAL pairs, as in the single-process
reference.  MoCo (with_moco) and stage 2 are 'next' rows and raise."""
import torch
from torch import nn

from antmmf.hip import contrastive
from .univl_video_base import UnivlVideoBase


class UnivlForVideoTextRetrieval(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if "stage2" in self.config.training_stage:
            raise NotImplementedError("stage2 (cross-encoder scoring + hard-negative mining): SURVEY.md 8(f) 'next' row")
        self.module = UnivlVideoBase(config, with_cross_encoder=False)
        self.with_moco = bool(self.config.get("with_moco", True))
        if self.with_moco:
            raise NotImplementedError("with_moco: true (MoCo queue + EMA key encoders): SURVEY.md 8(f) 'next' row; set with_moco: false")

    def get_l1_simi_matrix(self, text_embed_l1, video_embed_l1, num_clips):
        """[bsz_text, bsz_video, num_clips] clip-level scores (reference :199-226, cal_cross=True branch)."""
        d = video_embed_l1.size(-1)
        s = contrastive.matmul_f32(video_embed_l1.detach(), text_embed_l1.detach())  # [V*n, T]
        return s.view(-1, num_clips, s.shape[-1]).permute(2, 0, 1)

    def reduce_clips(self, simi_logits, level="l1"):
        return simi_logits.logsumexp(-1) if level == "l1" else simi_logits

    def forward_stage1(self, vis_input, cap_input, output_dict=None, cal_cross=True):
        output_dict = dict(losses={}) if output_dict is None else output_dict
        text_embed, video_embed, num_clips = cap_input[2], vis_input[2], vis_input[3]
        if self.training and cal_cross:
            loss = contrastive.mil_nce_sharded(text_embed, video_embed, num_clips)
        else:
            loss = text_embed.new_tensor(0.0, dtype=torch.float32)
        output_dict["losses"]["level1_similarity_loss"] = loss
        output_dict["l1_simi"] = self.reduce_clips(self.get_l1_simi_matrix(text_embed, video_embed, num_clips), "l1")
        return output_dict

    def forward_stage(self, cap_input, vis_input, cal_cross=True):
        return self.forward_stage1(vis_input, cap_input, None, cal_cross=cal_cross)

    def forward(self, img_input, caption_input, ocr_input=None, region_input=None, caption_output=None, sample_list=None):
        cap_input, vis_input, _, _ = self.module.get_l2_input(img_input, caption_input)
        return self.forward_stage(cap_input + (caption_input,), vis_input + (img_input,), True)

    def get_optimizer_parameters(self, config):
        """Four groups: {pretrained towers, new modules} x {decay, no decay} (reference :482-542)."""
        lr = config.optimizer_attributes.params.lr
        weight_decay = config.optimizer_attributes.params.weight_decay
        encoder_lr_decay = self.config.get("encoder_lr_decay", 0.01)
        no_decay = ("bias", "LayerNorm.bias", "LayerNorm.weight")
        tower_prefixes = ("text_encoder.embeddings.", "text_encoder.encoder.", "text_encoder.pooler.", "img_embeddings.", "img_encoder.")
        groups = {(t, dcy): [] for t in (True, False) for dcy in (True, False)}
        for n, p in self.named_parameters():
            tower = any(pre in n for pre in tower_prefixes)
            decays = not any(nd in n for nd in no_decay)
            groups[(tower, decays)].append(p)
        return [
            {"params": groups[(True, True)], "weight_decay": weight_decay, "lr": lr * encoder_lr_decay},
            {"params": groups[(False, True)], "weight_decay": weight_decay},
            {"params": groups[(True, False)], "weight_decay": 0.0, "lr": lr * encoder_lr_decay},
            {"params": groups[(False, False)], "weight_decay": 0.0},
        ]
