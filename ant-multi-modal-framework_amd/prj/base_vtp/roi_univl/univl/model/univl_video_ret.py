"""UnivlForVideoTextRetrieval, stage 1 (ITC / MIL-NCE) on the MI355X path (reference:
prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:16-31,146-226,251-387,445-542).

Training step: towers -> L2-normalised embeddings -> row-sharded global MIL-NCE (antmmf.hip.contrastive), which
replaces gather_tensor x2 + get_l1_simi_matrix + the tiled [T*n, V*n] matrix + get_mil_nce_loss.  `l1_simi`
(reported [T, V] scores, logsumexp over clips) is produced for the LOCAL pairs, as in the single-process
reference.  with_moco: true (the reference default) replaces the level-1 loss with the two-direction MoCo loss against the
momentum key encoders' queues (reference :262-312, moco_utils.py).

Stage 2 (reference :33-144,389-443): every (caption, video) pair of the local batch goes through the cross encoder
([text ; clips ; SEP] through the text tower's BERT layers) -> similarity_dense -> [T, V] scores -> MIL-NCE on that matrix;
optional hard-negative mining picks each caption's videos from the (gathered) level-1 scores.  The pair batch is built once per
chunk of caption rows (expand + one cat), there is no Python loop over captions in the mining branch, and the loss runs on
the fused row kernels."""
import torch
from torch import nn

from antmmf.hip import contrastive
from antmmf.hip import functional as HF
from antmmf.utils.distributed_utils import all_gather, gather_tensor, get_rank, get_world_size
from .moco_utils import MocoUtils
from .univl_video_base import UnivlVideoBase


class UnivlForVideoTextRetrieval(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        with_cross_encoder = "stage2" in self.config.training_stage
        self.module = UnivlVideoBase(config, with_cross_encoder=with_cross_encoder)
        if with_cross_encoder:
            self.dropout = nn.Dropout(0.1)
            self.similarity_dense = nn.Sequential(nn.Linear(self.config.hidden_size, self.config.hidden_size * 2), nn.ReLU(True),
                                                  nn.Linear(self.config.hidden_size * 2, 1))
        if "stage3" in self.config.training_stage:
            # DMAE head (reference: prj/dmae_vtp/roi_univl/univl/model/univl_video_ret.py:35-42,457-476); lives in the dmae_vtp overlay
            try:
                from .dmae_utils import CrossEn, DmaeUtils, NegNCE
            except ImportError as e:
                raise ImportError("training_stage with stage3 is the dmae_vtp project: put prj/dmae_vtp (not prj/base_vtp) on sys.path") from e
            self.module.need_cross_inputs = True
            self.dmae_utils = DmaeUtils(config)
            self.loss_type = self.config.get("l3_loss_type", "negNCE")
            assert self.loss_type in ["cross_entropy", "negNCE"]
            self.loss_fct = NegNCE() if self.loss_type == "negNCE" else CrossEn()
        self.pair_chunk_rows = int(self.config.get("cross_chunk_rows", 5))  # caption rows per cross-encoder call (reference: 5)
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

    # ------------------------------------------------------------------ stage 2
    def _score_pairs(self, cap_embed, cap_mask, vis_embed, vis_mask):
        """Cross-encode aligned pairs (row p of every argument) -> similarity_dense score [P]."""
        _, _, pooled = self.module.get_cross_output(cap_embed, vis_embed, cap_mask, vis_mask, 1)
        pooled = self.dropout(pooled) if self.training and self.dropout.p > 0 else pooled
        fc1, fc2 = self.similarity_dense[0], self.similarity_dense[2]
        hidden = HF.linear(pooled.contiguous(), fc1.weight, fc1.bias, act="relu")
        # the final Linear has ONE output: a [P, 2h] x [2h] product, done as an fp32 reduction (no GEMM tile to fill)
        return (hidden.float() * fc2.weight.float().view(1, -1)).sum(-1) + fc2.bias.float()

    def _cross_similarity(self, sequence_output, visual_output, attention_mask, video_mask, num_clips):
        """[b_text, b_visual] cross-encoder scores of all pairs, `pair_chunk_rows` caption rows per call (reference :33-89)."""
        b_text, b_visual = sequence_output.size(0), visual_output.size(0)
        rows = []
        for r0 in range(0, b_text, self.pair_chunk_rows):
            seq = sequence_output[r0:r0 + self.pair_chunk_rows]
            msk = attention_mask[r0:r0 + self.pair_chunk_rows]
            step = seq.size(0)
            seq_l = seq.unsqueeze(1).expand(-1, b_visual, -1, -1).reshape(step * b_visual, seq.size(1), seq.size(2))
            msk_l = msk.unsqueeze(1).expand(-1, b_visual, -1).reshape(step * b_visual, -1)
            vis_r = visual_output.unsqueeze(0).expand(step, -1, -1, -1).reshape(step * b_visual, visual_output.size(1), visual_output.size(2))
            vmk_r = video_mask.unsqueeze(0).expand(step, -1, -1).reshape(step * b_visual, -1)
            rows.append(self._score_pairs(seq_l, msk_l, vis_r, vmk_r).view(step, b_visual))
        return torch.cat(rows, dim=0)

    def _cross_similarity_hard_mining(self, vis_input, cap_input, l1_simi_matrix):
        """Each caption is scored against `bsz` videos chosen from the level-1 scores (top_k / nearliest), its own video forced
        onto the diagonal (reference :91-144) -- batched: one index tensor instead of a Python loop over captions."""
        sequence_output, attention_mask, bsz = cap_input[0], cap_input[1], cap_input[3]
        visual_output, video_mask = vis_input[0], vis_input[1]
        visual_output = gather_tensor(visual_output, method="cat", back_gradient=True, pad_tensors=True)
        video_mask = gather_tensor(video_mask, method="cat", back_gradient=True, pad_tensors=True)
        all_bsz = all_gather(bsz)
        beg_idx = sum(all_bsz[:get_rank()])
        own = torch.arange(bsz, device=l1_simi_matrix.device)
        raw = beg_idx + own
        score = l1_simi_matrix[raw].clone()                                    # [bsz, B_g]
        if self.config.re_sample_method == "top_k":
            score[own, raw] -= 100.0
            chosen = torch.topk(score, bsz, dim=1, sorted=False).indices
        elif self.config.re_sample_method == "nearliest":
            score = (score - score[own, raw].unsqueeze(1)).abs()
            score[own, raw] = 100.0
            chosen = torch.topk(score, bsz, dim=1, sorted=False, largest=False).indices
        else:
            raise ValueError(f"re_sample_method {self.config.re_sample_method!r}")
        chosen[own, own] = raw                                                 # the true pair sits on the diagonal
        self._last_chosen = chosen.detach()                                    # (for tests: topk(sorted=False) order is unspecified)
        flat = chosen.reshape(-1)
        vis_r, vmk_r = visual_output[flat], video_mask[flat]
        seq_l = sequence_output.unsqueeze(1).expand(-1, bsz, -1, -1).reshape(bsz * bsz, sequence_output.size(1), sequence_output.size(2))
        msk_l = attention_mask.unsqueeze(1).expand(-1, bsz, -1).reshape(bsz * bsz, -1)
        out = []
        step = self.pair_chunk_rows * bsz
        for p0 in range(0, bsz * bsz, step):
            out.append(self._score_pairs(seq_l[p0:p0 + step], msk_l[p0:p0 + step], vis_r[p0:p0 + step], vmk_r[p0:p0 + step]))
        return torch.cat(out).view(bsz, bsz)

    def get_l2_simi_matrix(self, cap_embed, cap_mask, visual_embed, visual_mask, num_clips, cal_cross=False):
        if cal_cross:
            return self._cross_similarity(cap_embed, visual_embed, cap_mask, visual_mask, num_clips)
        return self._score_pairs(*self.module._align_text_to_video_clips(cap_embed, cap_mask, 1), visual_embed, visual_mask).view(-1, 1)

    def scheduled_mining_draw(self, device):
        """The CN-VID schedule's coin (prj/cnvid_vtp/roi_univl/univl/model/univl_video_ret.py:410-420): every rank draws an integer in [0, 100), the MEAN over
        the ranks / 100 is compared with the trainer's `incre_num` -- so all ranks take the same branch (the two branches issue different collectives).
        One all-reduce of one float instead of the reference's padded list gather; the draw comes from torch's generator of `device` with the
        reference's own call, so a seeded single-process run takes the reference's decisions (tests/golden/e2e_cnvid_gate.pt)."""
        draw = torch.randint(low=0, high=100, size=[1], device=device, dtype=torch.float32)
        world = get_world_size()
        if world > 1:
            torch.distributed.all_reduce(draw)
        return float(draw) / world / 100.0

    def forward_stage2(self, vis_input, cap_input, output_dict=None, cal_cross=True, incre_num=None):
        """incre_num None: prj/base_vtp's head -- mine on every training step when `hard_example_mining` is set (univl_video_ret.py:389-401).  A number: the
        CN-VID head's scheduled gate (prj/cnvid_vtp/.../univl_video_ret.py:398-428) -- mine only when the rank-agreed draw falls below it; the row re-weighting
        applies on mined and plain steps alike, as there (:438-459)."""
        output_dict = dict(losses={}) if output_dict is None else output_dict
        cap_embed, cap_mask, batch_size = cap_input[0], cap_input[1], cap_input[3]
        visual_embed, visual_mask, num_clips = vis_input[0], vis_input[1], vis_input[3]
        configured = self.training and self.config.get("hard_example_mining", False)
        mining = configured
        if configured:
            l1_simi_clone = output_dict["l1_simi"].clone().detach()
            if incre_num is not None:
                mining = self.scheduled_mining_draw(l1_simi_clone.device) < float(incre_num)
        self._last_mined = bool(mining)
        if mining:
            l2_simi = self._cross_similarity_hard_mining(vis_input, cap_input, l1_simi_clone)
        else:
            l2_simi = self.get_l2_simi_matrix(cap_embed, cap_mask, visual_embed, visual_mask, num_clips, cal_cross=cal_cross)
        if cal_cross and l2_simi.size(0) == l2_simi.size(1):
            weight = None
            if configured and self.config.re_weight_method == "median":
                beg = sum(all_gather(batch_size)[:get_rank()])
                l1_diag = torch.diagonal(l1_simi_clone[beg:beg + batch_size, beg:beg + batch_size])
                l1_mean, l1_min = l1_diag.mean(), l1_diag.min()   # ("median" in the reference's config is a mean, :425)
                down = torch.clamp((l1_mean - l1_min) / (l1_diag - l1_min), min=0.2)
                weight = torch.where(l1_diag > l1_mean, down, torch.ones_like(l1_diag))
            loss = contrastive.mil_nce_matrix(l2_simi.view(batch_size, batch_size), weight)
        else:
            loss = l2_simi.new_tensor(0.0)
        output_dict["losses"]["level2_similarity_loss"] = loss
        output_dict["l2_simi"] = self.reduce_clips(l2_simi, "l2")
        return output_dict

    def forward_stage3(self, vis_input, cap_input, output_dict=None, cal_cross=True):
        """DMAE cross-modal retrieval head: token-wise interaction scores [T, V] -> CrossEn / NegNCE in both directions."""
        output_dict = dict(losses={}) if output_dict is None else output_dict
        l3_simi, margin_loss = self.dmae_utils.get_similarity_logits(vis_input, cap_input, shaped=True, loose_type=True)
        if cal_cross and l3_simi.size(0) == l3_simi.size(1):
            loss = (self.loss_fct(l3_simi) + self.loss_fct(l3_simi.t())) / 2
        else:
            loss = l3_simi.new_tensor(0.0)
        output_dict["losses"]["level3_similarity_loss"] = loss + margin_loss
        output_dict["l3_simi"] = self.reduce_clips(l3_simi, "l2")
        return output_dict

    def forward_stage(self, cap_input, vis_input, cal_cross=True, incre_num=None):
        output_dict = None
        if "stage1" in self.config.training_stage:
            output_dict = self.forward_stage1(vis_input, cap_input, output_dict, cal_cross=cal_cross)
        if "stage2" in self.config.training_stage:
            output_dict = self.forward_stage2(vis_input, cap_input, output_dict, cal_cross=cal_cross, incre_num=incre_num)
        if "stage3" in self.config.training_stage:
            output_dict = self.forward_stage3(vis_input, cap_input, output_dict, cal_cross=cal_cross)
        return output_dict

    def forward(self, img_input, caption_input, ocr_input=None, region_input=None, caption_output=None, sample_list=None):
        if sample_list is not None and "text_stage1_output" in sample_list and "visual_stage1_output" in sample_list:
            # retrieval evaluation: the towers ran once per batch, the block is scored from the cached stage-1 outputs (reference :466-472)
            cap_input, vis_input = tuple(sample_list["text_stage1_output"]), tuple(sample_list["visual_stage1_output"])
        else:
            cap_input, vis_input, _, _ = self.module.get_l2_input(img_input, caption_input)
        # CN-VID schedule: configs that carry `change_iter` make the trainer write `incre_num` into the batch (antmmf/trainers/base_trainer.py, reference
        # :552-571); the reference hands it to forward_stage from its pre-training head (prj/cnvid_vtp/.../univl_video_pretrain.py:182-188, default 0.0)
        incre_num = None
        if self.config.get("change_iter", None) is not None:
            incre_num = float(sample_list["incre_num"]) if sample_list is not None and "incre_num" in sample_list else 0.0
        return self.forward_stage(cap_input + (caption_input,), vis_input + (img_input,), True, incre_num=incre_num)

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
