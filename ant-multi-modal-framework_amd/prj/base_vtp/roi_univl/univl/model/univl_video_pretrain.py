"""Temporal head of UnivlForVideo (reference: prj/base_vtp/roi_univl/univl/model/univl_video_pretrain.py:60-90):
a learnable [cls] token + the clip features of a video through a small BERT (no word embeddings, no pooler) -- "config 3"
(n = 8 clips, 3 layers).  The reference builds that BERT with `TextEncoder(config.temporal_encoder)` of type
PretrainedTransformerEncoder (a HuggingFace AutoModel; transformers is not pinned, SURVEY.md 8c: parity unpinned at that
boundary); here the in-repo BERT (RobertBertEncoder family, fused HIP layers) is used, selected by the same config block.
The MLM / ITM pre-training heads of UnivlForVideo are outside the contrastive path and are not built."""
import torch
from torch import nn

from antmmf.modules.encoders import TextEncoder
from .univl_video_ret import UnivlForVideoTextRetrieval


class UnivlForVideo(UnivlForVideoTextRetrieval):
    def __init__(self, config):
        super().__init__(config)
        if self.config.get("with_temporal_encoder", False):
            self.add_temporal_head()

    def add_temporal_head(self):
        self.cls_token = nn.Parameter(torch.randn(1, 1, self.config.hidden_size))
        self.temporal_encoder = TextEncoder(self.config.temporal_encoder).module
        self.temporal_encoder.embeddings.word_embeddings = None
        if hasattr(self.temporal_encoder, "module"):
            self.temporal_encoder.module.pooler = None

    def get_temporal_output(self, clip_feat):
        """clip_feat [B, num_clips, hidden] -> sequence output [B, 1 + num_clips, hidden] (cls first)."""
        bsz, n_clips, _ = clip_feat.shape
        cls_tokens = self.cls_token.expand(bsz, -1, -1)
        input_embeds = torch.cat((cls_tokens.float(), clip_feat.float()), dim=1)
        enc = self.temporal_encoder
        x = enc.embeddings(inputs_embeds=input_embeds)
        key_bias = torch.zeros((bsz, n_clips + 1), dtype=torch.float32, device=clip_feat.device)  # every position attended
        return enc.encoder(x, key_bias, head_mask=None)[0]
