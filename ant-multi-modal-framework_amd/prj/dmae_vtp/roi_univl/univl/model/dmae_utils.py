"""DMAE retrieval head pieces on the MI355X path (reference: prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py).

Built so far (SURVEY.md section 8a rows T11b and the L5 losses):
  * LayerNormDmae / ResidualAttentionBlockDmae / TransformerClip (:574-619) -- CLIP4Clip's temporal transformer: the fused
    HIP transformer layer, kind "clip", LayerNorm eps 1e-12, additive key mask;
  * DmaeUtils._agg_visual_feat (:186-227), meanP and seqTransf;
  * CrossEn (:528-537) and NegNCE (:539-563) on fused row kernels.
Not built yet: the WTI token-wise interaction (_get_wti_similarity / wti_interaction, :85-184) and TPM-CL
(get_partial_similarity, :280-463) raise NotImplementedError."""
from collections import OrderedDict

import torch
from torch import nn

from antmmf.hip import contrastive
from antmmf.hip import functional as HF
from antmmf.modules.vision.backbone.clip.model import QuickGELU


class LayerNormDmae(nn.Module):
    """TF-style LayerNorm (epsilon inside the square root) -- parameter holder; the arithmetic runs inside the fused layer."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return HF.layer_norm(x, self.weight, self.bias, self.variance_epsilon)


class ResidualAttentionBlockDmae(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        if d_model // n_head != 64:
            raise ValueError("the fused attention kernel is specialised for head_dim 64 (transformer_heads = width // 64 in the reference)")
        self.attn = nn.MultiheadAttention(d_model, n_head)  # parameter holder (same names / init as the reference)
        self.ln_1 = LayerNormDmae(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNormDmae(d_model)
        self.n_head = n_head
        self._spec = HF.LayerSpec(kind="clip", heads=n_head, eps=self.ln_1.variance_epsilon, act="quick_gelu", packed_qkv=True)

    def _params(self):
        return dict(ln1_w=self.ln_1.weight, ln1_b=self.ln_1.bias, wqkv=self.attn.in_proj_weight, bqkv=self.attn.in_proj_bias,
                    wo=self.attn.out_proj.weight, bo=self.attn.out_proj.bias, ln2_w=self.ln_2.weight, ln2_b=self.ln_2.bias,
                    w1=self.mlp.c_fc.weight, b1=self.mlp.c_fc.bias, w2=self.mlp.c_proj.weight, b2=self.mlp.c_proj.bias)

    def forward(self, para_tuple: tuple):
        """(x [B, N, d] bf16, key_bias [B, N] fp32 additive) -> same tuple (the reference threads (x, attn_mask) the same way)."""
        x, key_bias = para_tuple
        return HF.transformer_layer(x, self._spec, self._params(), key_bias=key_bias), key_bias


class TransformerClip(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlockDmae(width, heads) for _ in range(layers)])

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor):
        """Reference calling convention: x is LND, attn_mask [B, L, L] additive.  The mask DmaeUtils builds is constant along
        the query axis ((1 - video_mask) * -1e6 expanded, :205-206); row 0 is taken as the per-key bias of the fused kernel."""
        key_bias = attn_mask[:, 0, :].float().contiguous()
        y, _ = self.resblocks((x.permute(1, 0, 2).contiguous().to(torch.bfloat16), key_bias))
        return y.permute(1, 0, 2)


class CrossEn(nn.Module):
    def forward(self, sim_matrix, logit_scale=100.0):
        return contrastive.cross_en(sim_matrix, logit_scale)


class NegNCE(nn.Module):
    def __init__(self):
        super().__init__()
        self.c_pos_w = 1.0
        self.c_neg_w = 0.5
        self.margin = 0.0

    def forward(self, sim_matrix, logit_scale=100.0):
        return contrastive.neg_nce(sim_matrix, logit_scale, self.c_pos_w, self.c_neg_w, self.margin)


class DmaeUtils(nn.Module):
    def __init__(self, config=dict()):
        super().__init__()
        self.config = config
        g = config.get
        self.interaction = g("l3_interaction", "wti")
        self.with_va = g("l3_with_nfc", True)
        self.wti_arch = g("l3_wti_arch", 1)
        self.sim_header = g("l3_sim_header", "meanP")
        self.partial_type = g("l3_partial_type", 4)
        self.max_frames = g("l3_max_frames", 8)
        self.max_words = g("l3_max_words", 30)
        self.cross_num_hidden_layers = g("l3_sim_header_hidden_layer", 4)
        hidden_size = g("hidden_size", 768)
        assert self.sim_header in ["meanP", "seqTransf"]
        if self.partial_type > 0:
            raise NotImplementedError("TPM-CL partial-order loss (l3_partial_type > 0): SURVEY.md 8a row L6, not built; set l3_partial_type: -1")
        if "wti" in self.interaction:
            if self.wti_arch != 1:
                raise NotImplementedError("l3_wti_arch 2 / 3 (MLP weight heads)")
            self.text_weight_fc = nn.Linear(hidden_size, 1)
            self.video_weight_fc = nn.Linear(hidden_size, 1)
        if self.sim_header == "seqTransf":
            self.frame_position_embeddings = nn.Embedding(77, hidden_size)
            self.transformerClip = TransformerClip(width=hidden_size, layers=self.cross_num_hidden_layers, heads=hidden_size // 64)

    def _agg_visual_feat(self, visual_output, video_mask, sim_header="meanP"):
        """[B, n*e, d] frame tokens + [B, n] mask -> (aggregated tokens, token mask, original tokens), one token per frame."""
        expand_times = visual_output.shape[1] // video_mask.shape[1]
        video_token_mask = video_mask.unsqueeze(1).repeat(1, 1, expand_times).view(video_mask.shape[0], -1)
        visual_output_original = visual_output
        if sim_header == "seqTransf":
            seq_length = visual_output.size(1)
            pos = self.frame_position_embeddings.weight[:seq_length]
            x = visual_output + pos.to(visual_output.dtype)[None]
            extended_video_mask = ((1.0 - video_token_mask.float().unsqueeze(1)) * -1000000.0).expand(-1, seq_length, -1)
            x = self.transformerClip(x.permute(1, 0, 2), extended_video_mask).permute(1, 0, 2)
            visual_output = x.to(visual_output_original.dtype) + visual_output_original
        idx = torch.arange(0, visual_output.shape[1], expand_times, dtype=torch.long, device=visual_output.device)
        return visual_output[:, idx, :], video_token_mask[:, idx], visual_output_original[:, idx, :]

    def wti_interaction(self, *args, **kwargs):
        raise NotImplementedError("WTI token-wise interaction: SURVEY.md 8a row L5 (similarity), not built in this round")
