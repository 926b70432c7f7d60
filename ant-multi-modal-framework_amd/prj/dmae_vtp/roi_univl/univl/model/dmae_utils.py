"""DMAE retrieval head pieces on the MI355X path (reference: prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py).

Built:
  * LayerNormDmae / ResidualAttentionBlockDmae / TransformerClip (:574-619) -- CLIP4Clip's temporal transformer: the fused
    HIP transformer layer, kind "clip", LayerNorm eps 1e-12, additive key mask;
  * DmaeUtils._agg_visual_feat (:186-227), meanP and seqTransf;
  * the weighted token-wise interaction (_get_wti_similarity / wti_interaction / _loose_similarity / get_similarity_logits,
    :85-184,229-278): split GEMM -> [A*T, B*V] slab -> one fused max / arg-max reduction kernel (antmmf_wti_reduce_*), in blocks of
    text rows -- the reference's [B_t, B_v, N_t, N_v] einsum tensor (96 GB at B_g = 8192) is never materialised;
  * CrossEn (:528-537) and NegNCE (:539-563) on fused row kernels.
Not built: TPM-CL (get_partial_similarity, :280-463; l3_partial_type > 0 raises NotImplementedError) and wti_arch 2 / 3."""
from collections import OrderedDict

import torch
from torch import nn

from antmmf.hip import contrastive
from antmmf.hip import functional as HF
from antmmf.modules.vision.backbone.clip.model import QuickGELU
from antmmf.utils.distributed_utils import gather_tensor, get_world_size


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

    @staticmethod
    def _masked_softmax(fc, feat, mask):
        z = torch.nn.functional.linear(feat.float(), fc.weight.float(), fc.bias.float()).squeeze(2)   # Linear(D, 1): a [*, D] x [D] reduction
        return torch.softmax(z.masked_fill(mask.float() < 0.5, float("-inf")), dim=-1)

    def _get_wti_similarity(self, text_feat, video_feat, text_mask, video_mask, text_weight=None, video_weight=None, self_weight=False):
        return contrastive.wti_similarity(text_feat, video_feat, text_mask, video_mask, text_weight, video_weight,
                                          self_weight=self_weight, weighted="wti" in self.interaction)

    def wti_interaction(self, text_feat, word_feat, video_feat, word_mask, video_mask):
        """text_feat [A, 1, D] sentence embedding, word_feat [A, Nw, D] or None, video_feat [B, V, D]; masks 1 = real.
        Multi-GPU training: like the reference (:135-146) every rank scores the GLOBAL batch -- features and masks are all-gathered
        with gradient (gather_tensor: backward = reduce-sum to the owner, i.e. W x the single-process gradient before the
        data-parallel mean) -- but without the reference's forced barrier and its `torch.cuda.is_available()` gate."""
        if self.training and get_world_size() > 1:
            text_feat = gather_tensor(text_feat.contiguous(), method="cat", back_gradient=True, pad_tensors=True)
            if word_feat is not None:
                word_feat = gather_tensor(word_feat.contiguous(), method="cat", back_gradient=True, pad_tensors=True)
            video_feat = gather_tensor(video_feat.contiguous(), method="cat", back_gradient=True, pad_tensors=True)
            word_mask = gather_tensor(word_mask.float().contiguous(), method="cat", back_gradient=False, pad_tensors=True)
            video_mask = gather_tensor(video_mask.float().contiguous(), method="cat", back_gradient=False, pad_tensors=True)
        expand_times = video_feat.shape[1] // video_mask.shape[1]
        video_mask = video_mask.unsqueeze(1).repeat(1, 1, expand_times).view(video_mask.shape[0], -1)
        text_mask = word_mask
        if word_mask.shape[1] != text_feat.shape[1]:
            text_mask = word_mask[:, 0].reshape(text_feat.shape[0], -1)
        text_weight = word_weight = video_weight = None
        if "wti" in self.interaction:
            text_weight = self._masked_softmax(self.text_weight_fc, text_feat, text_mask)
            if word_feat is not None:
                word_weight = self._masked_softmax(self.text_weight_fc, word_feat, word_mask)
            video_weight = self._masked_softmax(self.video_weight_fc, video_feat, video_mask)
        logits = self._get_wti_similarity(text_feat, video_feat, text_mask, video_mask, text_weight, video_weight, self_weight=self.with_va)
        if self.interaction in ["att_ti", "att_wti"] and word_feat is not None:
            words = self._get_wti_similarity(word_feat, video_feat, word_mask, video_mask, word_weight, video_weight, self_weight=self.with_va)
            logits = (logits + words) / 2.0
        return logits

    def _loose_similarity(self, sequence_output, visual_output, attention_mask, video_mask, sim_header="meanP"):
        sequence_token_hidden = None
        if isinstance(sequence_output, tuple):
            sequence_output, sequence_token_hidden = sequence_output
        agg, video_token_mask, _ = self._agg_visual_feat(visual_output.contiguous(), video_mask, sim_header=sim_header)
        if "ti" not in self.interaction:
            raise NotImplementedError(f"l3_interaction {self.interaction!r}")
        return self.wti_interaction(sequence_output.contiguous(), sequence_token_hidden, agg, attention_mask, video_token_mask)

    def get_similarity_logits(self, vis_input, cap_input, output_dict=None, shaped=False, loose_type=False):
        """(simi_matrix [B_t, B_v], margin_loss) of stage 3 (reference :249-278); TPM-CL's margin loss is not built (0)."""
        cap_embed, cap_mask, text_embed_l1, _, twm_cap_mask = cap_input[:5]
        visual_embed, visual_mask = vis_input[0], vis_input[1]
        cap_embed = cap_embed.float() / cap_embed.float().norm(dim=-1, keepdim=True)
        visual_embed = visual_embed.float() / visual_embed.float().norm(dim=-1, keepdim=True)
        if twm_cap_mask is not None:
            cap_mask = twm_cap_mask
        if not shaped:
            cap_mask = cap_mask.view(-1, cap_mask.shape[-1])
            visual_mask = visual_mask.view(-1, visual_mask.shape[-1])
        if not loose_type:
            raise NotImplementedError("tight similarity header")
        simi = self._loose_similarity((text_embed_l1.float().unsqueeze(1), cap_embed), visual_embed, cap_mask, visual_mask, sim_header=self.sim_header)
        return simi, 0.0
