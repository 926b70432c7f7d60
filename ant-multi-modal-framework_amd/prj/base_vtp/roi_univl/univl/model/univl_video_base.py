"""UnivlVideoBase: the two towers + pooling + L2 normalisation of the contrastive step (reference:
prj/base_vtp/roi_univl/univl/model/univl_video_base.py:14-166,272-299).  In scope: arch_type "clip", stage 1
(ITC).  The stage-2 cross encoder (prepare_cross_*, get_cross_output; reference :168-271) is a "next" row of
SURVEY.md section 8(f) and raises here."""
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
        if self.arch_type != "clip":
            raise NotImplementedError("HIP path: arch_type 'clip' (in-repo ViT + BERT); the HF-AutoModel 'univl' arch is out of scope")
        self.build()

    def build(self):
        self.text_encoder = TextEncoder(self.config.text_encoder).module
        self.img_encoder = VisualEncoder(self.config.image_encoder).module
        self.img_proj = None
        if self.img_encoder.out_dim != self.config.hidden_size:
            self.img_proj = nn.Parameter(torch.empty((self.img_encoder.out_dim, self.config.hidden_size)))
            nn.init.normal_(self.img_proj, std=self.img_encoder.out_dim ** -0.5)
        # the cross encoder shares the text tower's embeddings / layers (reference :47-48)
        self.cross_embeddings = self.text_encoder.embeddings
        self.cross_encoder = self.text_encoder.encoder

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
        clip_feature = HF.l2_normalize(clip_feature.to(feat.dtype).contiguous())
        return dict(visual_embed=clip_tokens, visual_mask=clip_mask, visual_grid_shape=grid_feature.shape[-2:],
                    clip_feature=clip_feature)

    def forward_text_encoder(self, input_ids, input_mask, txt_encoder=None):
        text_encoder = txt_encoder or self.text_encoder
        sequence_output, pooled_output = text_encoder(input_ids=input_ids, attention_mask=input_mask)
        pooled_output = HF.l2_normalize(pooled_output.contiguous())
        return dict(sequence_output=sequence_output, pooled_output=pooled_output, input_mask=input_mask, words_importance=None)

    def get_l2_input(self, img_input, caption_input):
        visual = self.forward_img_encoder(**img_input)
        text = self.forward_text_encoder(caption_input["caption_raw_input_ids"], caption_input["caption_input_mask"])
        n_clips = visual["visual_embed"].shape[1]
        batch_size = text["pooled_output"].shape[0]
        # (cap_embed, cap_mask) / (visual_embed, visual_mask) feed only the stage-2 cross encoder; stage 1 carries None
        cap_input = (None, caption_input["caption_input_mask"], text["pooled_output"], batch_size)
        vis_input = (visual["visual_embed"], visual["visual_mask"], visual["clip_feature"], n_clips)
        return cap_input, vis_input, text, visual

    def get_cross_output(self, *args, **kwargs):
        raise NotImplementedError("stage-2 cross encoder: SURVEY.md section 8(f) 'next' row, not built in this round")
