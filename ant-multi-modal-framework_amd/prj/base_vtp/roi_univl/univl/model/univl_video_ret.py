"""UnivlForVideoTextRetrieval, stage 1 (ITC / MIL-NCE) on the MI355X path (reference:
prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:16-31,146-226,251-387,445-542).

Training step: towers -> L2-normalised embeddings -> row-sharded global MIL-NCE (antmmf.hip.contrastive), which
replaces gather_tensor x2 + get_l1_simi_matrix + the tiled [T*n, V*n] matrix + get_mil_nce_loss.  `l1_simi`
(reported [T, V] scores, logsumexp over clips) is produced for the LOCAL pairs, as in the single-process
reference.  with_moco: true (the reference default) replaces the level-1 loss with the two-direction MoCo loss against the
momentum key encoders' queues (reference :262-312, moco_utils.py).  Stage 2 is a 'next' row and raises."""
import torch
from torch import nn

from antmmf.hip import contrastive
from .moco_utils import MocoUtils
from .univl_video_base import UnivlVideoBase


class UnivlForVideoTextRetrieval(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if "stage2" in self.config.training_stage:
            raise NotImplementedError("stage2 (cross-encoder scoring + hard-negative mining): SURVEY.md 8(f) 'next' row")
        self.module = UnivlVideoBase(config, with_cross_encoder=False)
        self.with_moco = bool(self.config.get("with_moco", True))
        self.moco_utils = None  # built lazily at the first training step, as in the reference (:263-268)

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
        if self.training and self.with_moco:
            loss = self.get_moco_loss(vis_input, cap_input)
        elif self.training and cal_cross:
            loss = contrastive.mil_nce_sharded(text_embed, video_embed, num_clips)
        else:
            loss = text_embed.new_tensor(0.0, dtype=torch.float32)
        output_dict["losses"]["level1_similarity_loss"] = loss
        output_dict["l1_simi"] = self.reduce_clips(self.get_l1_simi_matrix(text_embed, video_embed, num_clips), "l1")
        return output_dict

    def get_moco_loss(self, vis_input, cap_input):
        """Level-1 loss with MoCo (reference get_simi_logits :262-312): momentum-update the key towers, encode keys without
        gradient, score  q_video vs (k_text+, text queue)  and  q_text vs (k_video+ clips, video queue), average, enqueue."""
        text_embed, video_embed, num_clips = cap_input[2], vis_input[2], vis_input[3]
        caption_input, img_input = cap_input[-1], vis_input[-1]
        if self.moco_utils is None:
            self.moco_utils = MocoUtils(self.config, img_encoder=self.module.img_encoder, txt_encoder=self.module.text_encoder).to(text_embed.device)
        mu = self.moco_utils
        with torch.no_grad():
            mu.momentum_update_key_encoder()
            key_v = self.module.forward_img_encoder(**img_input, img_encoder=mu.img_encoder_k)["clip_feature"]      # [B*n, D]
            key_t = self.module.forward_text_encoder(caption_input["caption_raw_input_ids"], caption_input["caption_input_mask"],
                                                     txt_encoder=mu.txt_encoder_k)["pooled_output"]                 # [B, D]
        d = key_t.shape[-1]
        # 1. q = clips, k+ = the caption's key, k- = text queue
        kpos_t = key_t.float().repeat_interleave(num_clips, 0).view(-1, 1, d)
        loss_v = mu.moco_loss(video_embed, kpos_t, mu.txt_queue)
        # 2. q = captions, k+ = the video's clip keys, k- = video queue
        loss_t = mu.moco_loss(text_embed, key_v.float().view(-1, num_clips, d), mu.img_queue)
        mu.dequeue_and_enqueue(key_v, key_t)
        return (loss_t + loss_v) / 2.0

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
