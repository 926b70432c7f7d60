"""UnivlVideoBase: the two towers + pooling + L2 normalisation of the contrastive step (reference:
prj/base_vtp/roi_univl/univl/model/univl_video_base.py:14-166,272-299) and the stage-2 cross-modal merged attention
(prepare_cross_text / prepare_cross_visual / get_cross_output, reference :168-271): [text tokens ; clip tokens ; SEP] through
the text tower's own BERT layers (fused HIP layers with the -10000 key mask), pooled = cls @ text_projection.
arch_type "clip" (in-repo CLIP ViT + BERT) or "univl" (PretrainedTransformerEncoder: HF-BERT weight layout on the fused BERT, pooler heads, img_fc)."""
import torch
from torch import nn

from antmmf.hip import functional as HF
from antmmf.modules.encoders import TextEncoder, VisualEncoder


class UnivlVideoBase(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        self.arch_type = self.config.get("arch_type", "univl")
        self.with_cross_encoder = kwargs.get("with_cross_encoder", None)
        if self.with_cross_encoder is None:
            self.with_cross_encoder = self.config.with_cross_encoder
        if self.arch_type not in ("clip", "univl"):
            raise NotImplementedError(f"arch_type {self.arch_type!r}: 'clip' (in-repo ViT + BERT towers) or 'univl' (PretrainedTransformerEncoder text tower)")
        self.build()

    def build(self):
        self.text_encoder = TextEncoder(self.config.text_encoder).module
        self.img_encoder = VisualEncoder(self.config.image_encoder).module
        self.img_proj = None
        if self.img_encoder.out_dim != self.config.hidden_size:
            self.img_proj = nn.Parameter(torch.empty((self.img_encoder.out_dim, self.config.hidden_size)))
            nn.init.normal_(self.img_proj, std=self.img_encoder.out_dim ** -0.5)
        if self.arch_type == "univl":   # reference :37-45: an MLP on the clip feature, registered on the image encoder
            h = self.config.hidden_size
            self.img_encoder.add_module("img_fc", nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, h)))
        # the cross encoder shares the text tower's embeddings / layers (reference :47-48)
        self.cross_embeddings = self.text_encoder.embeddings
        self.cross_encoder = self.text_encoder.encoder
        if self.with_cross_encoder is True and self.arch_type == "univl":
            import copy

            self.cross_pooler = copy.deepcopy(self.text_encoder.pooler)   # its own parameters (reference :51-54)

    def forward_img_encoder(self, image_data, image_pad_mask, image_n_clips, image_num_frames, img_encoder=None, **kwargs):
        img_encoder = img_encoder or self.img_encoder
        out = img_encoder(image_data, image_mask=image_pad_mask)
        grid_feature, grid_mask = out["grid_feature"], out["grid_mask"]  # [b, T, c, h, w], [b, T, h, w]
        n_clips, n_frames = int(image_n_clips[0]), int(image_num_frames[0])
        bsz, c = grid_feature.size(0), grid_feature.size(2)
        feat = grid_feature.reshape(bsz * n_clips, n_frames, c, -1)  # [b*n, f, c, hw]
        if self.img_proj is not None:
            feat = HF.linear(feat.transpose(2, 3).contiguous(), self.img_proj, weight_layout="io").transpose(2, 3)
            c = feat.size(2)
        keep = (~grid_mask.reshape(bsz * n_clips, n_frames, 1, -1)).to(torch.float32)
        # clip feature = masked mean over the clip's frames and grid cells (reference :93-96); tiny [b*n, f, c] tensor
        clip_feature = (feat.float() * keep).sum(dim=(1, 3)) / keep.sum(dim=(1, 3))
        clip_tokens = clip_feature.view(bsz, n_clips, c)
        clip_mask = torch.zeros((bsz, n_clips), device=clip_tokens.device, dtype=torch.bool)
        if "img_fc" in img_encoder._modules:
            fc = img_encoder.img_fc
            hid = HF.linear(clip_feature.to(torch.bfloat16).contiguous(), fc[0].weight, fc[0].bias, act="relu")
            clip_feature = HF.linear(hid, fc[2].weight, fc[2].bias)
        clip_feature = HF.l2_normalize(clip_feature.to(feat.dtype).contiguous())
        return dict(visual_embed=clip_tokens, visual_mask=clip_mask, visual_grid_shape=grid_feature.shape[-2:],
                    clip_feature=clip_feature)

    def forward_text_encoder(self, input_ids, input_mask, txt_encoder=None):
        text_encoder = txt_encoder or self.text_encoder
        words_importance = None
        if self.arch_type == "univl":
            # all-zero token types (reference :125): None spares BertEmbeddings its any() host sync.  In training the reference asks for the attention maps and
            # reduces them to `words_importance` (:131-143): here the tower hands back that reduction itself (modeling_bert.KeyImportance)
            # -- ONLY when something will read it: `words_importance` costs one more score pass per BERT layer and its one reader on the reference side is the
            # pre-training head's attentive masking (univl_video_pretrain.py:194), which is outside this build.  A head that wants it sets
            # `model.module.want_words_importance = True` (or the config key `words_importance: true`); otherwise the entry stays None
            want = bool(self.training) and bool(getattr(self, "want_words_importance", False) or self.config.get("words_importance", False))
            out = text_encoder(input_ids=input_ids, attention_mask=input_mask, token_type_ids=None, output_attentions=want)
            sequence_output, pooled_output = out[0], out[1]
            if want:
                words_importance = out[2].value.detach()
        else:
            sequence_output, pooled_output = text_encoder(input_ids=input_ids, attention_mask=input_mask)
        pooled_output = HF.l2_normalize(pooled_output.contiguous())
        return dict(sequence_output=sequence_output, pooled_output=pooled_output, input_mask=input_mask, words_importance=words_importance)

    # ------------------------------------------------------------------ stage-2 cross encoder
    def prepare_cross_text(self, input_ids, input_mask):
        cap_embed = self.cross_embeddings(input_ids=input_ids, token_type_ids=None)   # type 0 everywhere (reference :171-176)
        return cap_embed, input_mask, cap_embed.shape[0]

    def prepare_cross_visual(self, visual_embed, visual_mask=None):
        """clip tokens + the [SEP] (id 102) word embedding, token type 1, positions 0..n  (reference :178-204)."""
        bsz, num_clip = visual_embed.shape[0], visual_embed.shape[1]
        if visual_mask is None:
            visual_mask = torch.zeros((bsz, num_clip), device=visual_embed.device).bool()
        sep_id = torch.full((bsz,), 102, dtype=torch.long, device=visual_embed.device)
        sep = self.cross_embeddings.word_embeddings(sep_id).unsqueeze(1)
        visual_mask = visual_mask.logical_not().long()
        inputs_embeds = torch.cat([visual_embed.float(), sep.float()], 1)
        token_type_ids = torch.ones(inputs_embeds.shape[:2], dtype=torch.long, device=inputs_embeds.device)
        new_visual_embed = self.cross_embeddings(inputs_embeds=inputs_embeds, token_type_ids=token_type_ids)
        new_visual_mask = torch.cat([visual_mask, visual_mask.new_ones((bsz, 1))], 1)
        return new_visual_embed, new_visual_mask, num_clip

    def build_transformer_input(self, visual_embed_dict, text_embed_dict, caption_input):
        cap_embed, cap_mask, batch_size = self.prepare_cross_text(caption_input["caption_input_ids"], caption_input["caption_input_mask"])
        visual_embed, visual_mask, num_clip = self.prepare_cross_visual(visual_embed_dict["visual_embed"], visual_embed_dict["visual_mask"])
        return cap_embed, visual_embed, cap_mask, visual_mask, num_clip, batch_size

    def _align_text_to_video_clips(self, cap_embed, cap_mask, num_clip: int = 1):
        if num_clip > 1:
            cap_embed = cap_embed.repeat_interleave(num_clip, dim=0)
            cap_mask = cap_mask.repeat_interleave(num_clip, dim=0)
        return cap_embed, cap_mask

    def get_cross_output(self, cap_embed, visual_embed, cap_mask, visual_mask, n_clips):
        """-> (text part, visual part without its SEP, pooled = cls @ text_projection)  (reference :224-271)."""
        cap_embed, cap_mask = self._align_text_to_video_clips(cap_embed, cap_mask, n_clips)
        n_text = cap_embed.size(1)
        embed = torch.cat([cap_embed, visual_embed], 1).contiguous()
        mask = torch.cat([cap_mask, visual_mask], 1)
        key_bias = (1.0 - mask.float()) * -10000.0
        sequence_output = self.cross_encoder(embed, key_bias.contiguous(), head_mask=None)[0]
        cls = sequence_output[:, 0, :].contiguous()
        if self.arch_type == "univl":
            pooled_output = self.cross_pooler(sequence_output).to(sequence_output.dtype)   # (the pooler's tanh runs in fp32; the similarity MLP takes the activation dtype)
        elif self.text_encoder.text_projection is not None:
            pooled_output = HF.linear(cls, self.text_encoder.text_projection, weight_layout="io")
        else:
            pooled_output = cls
        return sequence_output[:, :n_text], sequence_output[:, n_text:-1], pooled_output

    def get_l2_input(self, img_input, caption_input):
        visual = self.forward_img_encoder(**img_input)
        text = self.forward_text_encoder(caption_input["caption_raw_input_ids"], caption_input["caption_input_mask"])
        n_clips = visual["visual_embed"].shape[1]
        batch_size = text["pooled_output"].shape[0]
        twm_input_mask = caption_input.get("caption_twm_input_mask")  # DMAE's token-weighting mask (dmae_vtp :293-300), usually absent
        if self.with_cross_encoder or getattr(self, "need_cross_inputs", False):
            cap_embed, visual_embed, cap_mask, visual_mask, n_clips, batch_size = self.build_transformer_input(visual, text, caption_input)
            cap_input = (cap_embed, cap_mask, text["pooled_output"], batch_size, twm_input_mask)
            vis_input = (visual_embed, visual_mask, visual["clip_feature"], n_clips)
        else:
            # (cap_embed, cap_mask) / (visual_embed, visual_mask) feed only the stage-2 / stage-3 heads; stage 1 carries None
            cap_input = (None, caption_input["caption_input_mask"], text["pooled_output"], batch_size, twm_input_mask)
            vis_input = (visual["visual_embed"], visual["visual_mask"], visual["clip_feature"], n_clips)
        return cap_input, vis_input, text, visual

    def forward(self, img_input, caption_input):
        cap_input, vis_input, _, _ = self.get_l2_input(img_input, caption_input)
        return self.get_cross_output(cap_input[0], vis_input[0], cap_input[1], vis_input[1], vis_input[3])
