"""DMAE retrieval head pieces on the MI355X path (reference: prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py).

Built:
  * LayerNormDmae / ResidualAttentionBlockDmae / TransformerClip (:574-619) -- CLIP4Clip's temporal transformer: the fused
    HIP transformer layer, kind "clip", LayerNorm eps 1e-12, additive key mask;
  * DmaeUtils._agg_visual_feat (:186-227), meanP and seqTransf;
  * the weighted token-wise interaction (_get_wti_similarity / wti_interaction / _loose_similarity / get_similarity_logits,
    :85-184,229-278): split GEMM -> [A*T, B*V] slab -> one fused max / arg-max reduction kernel (antmmf_wti_reduce_*), in blocks of
    text rows -- the reference's [B_t, B_v, N_t, N_v] einsum tensor (96 GB at B_g = 8192) is never materialised;
  * CrossEn (:528-537) and NegNCE (:539-563) on fused row kernels.
  * TPM-CL, the partial-order margin losses (get_partial_similarity / _get_partial_output / wti_interaction_row, :280-523, with
    tpmcl_utils.py's LinearXWeightPredictor and TokenImportanceSelector): a small head over 8 x 16 caption/video blocks, composed
    from torch device ops (no custom kernel: per block it touches ~128 pairs x 30 tokens; the towers dominate the step).
Not built: the attention-based predictor variant (xwp_type "attention"; the reference hard-codes "linear")."""
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


class LinearXWeightPredictor(nn.Module):
    """Token-importance predictor (reference: tpmcl_utils.py:6-50): the query tokens are mapped along their token axis onto the key's
    token count, concatenated with the key tokens, LayerNorm over the [tokens, 2D] slab, a 2-layer MLP and a sigmoid; weights are
    normalised to sum 1 over the tokens.  Same parameter names as the reference (q_proj / k_proj are only used when the input widths
    differ from embed_dim, which never happens on this path)."""

    def __init__(self, num_frames: int, num_tokens: int, embed_dim: int, qk_bias: bool = False, qdim: int = None, kdim: int = None):
        super().__init__()
        self.num_frames, self.num_tokens, self.embed_dim = num_frames, num_tokens, embed_dim
        self.qdim = qdim if qdim is not None else embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self._qk_same_embed_dim = self.qdim == embed_dim and self.kdim == embed_dim
        self.q_proj = nn.Linear(self.qdim, embed_dim, bias=qk_bias)
        self.k_proj = nn.Linear(self.kdim, embed_dim, bias=qk_bias)
        self.qk_proj = nn.Linear(self.num_frames, self.num_tokens, bias=qk_bias)
        self.attn_proj = nn.Sequential(nn.LayerNorm([num_tokens, embed_dim * 2]), nn.Linear(embed_dim * 2, embed_dim // 2, bias=False), nn.GELU(),
                                       nn.Linear(embed_dim // 2, 1, bias=False), nn.Sigmoid())

    # rows (pairs x tokens) from which the 2D -> D/2 Linear of the MLP runs on the bf16 MFMA GEMM (fp32 accumulation) instead of torch's fp32 matmul:
    # with all 8 x 16 blocks of a step batched it is 0.26 TFLOP per predictor and direction, and a rocBLAS fp32 GEMM on it was 5 % of the dmae12 step.
    # (The reference applies .float() to the inputs, but under its autocast context F.linear runs in bf16 there as well.)  None: never.
    MFMA_MIN_ROWS = 8192

    def forward(self, q, k):
        if not self._qk_same_embed_dim:
            q, k = self.q_proj(q), self.k_proj(k)
        q = self.qk_proj(q.float().transpose(-2, -1)).transpose(-1, -2)
        x = torch.cat([q, k.float()], dim=-1)
        rows = x.shape[0] * x.shape[1]
        if self.MFMA_MIN_ROWS is not None and rows >= self.MFMA_MIN_ROWS and x.shape[-1] % 64 == 0:
            ln, fc_a, act, fc_b, sig = self.attn_proj
            h = HF.linear(ln(x).to(torch.bfloat16), fc_a.weight)
            w = sig(fc_b(act(h.float()))).squeeze(-1)
        else:
            w = self.attn_proj(x).squeeze(-1)
        return w / w.sum(dim=1, keepdim=True)


    def forward_all_pairs(self, q_items, k_items):
        """forward(q, k) for EVERY (k item, q item) pair without materialising the pair batch: -> [n_k, n_q, tokens].

        The slab of a pair is cat(q'_j, k_i) with q' = qk_proj(q_j) depending on the q item only and k_i on the k item only, so
          * its LayerNorm statistics are sums of per-item sums:  mu_ij = (s1q_j + s1k_i) / N,  E[x^2]_ij = (s2q_j + s2k_i) / N;
          * the first Linear of the MLP is linear in the slab:  h_ij = r_ij (A_j + B_i) - r_ij mu_ij c + d  with
            A_j = (gamma_q * q'_j) W_q^T,  B_i = (gamma_k * k_i) W_k^T  (one small GEMM per ITEM, not per pair),  c = gamma W^T,  d = beta W^T.
        What is left per pair is the elementwise tail (GELU, the D/2 -> 1 dot, sigmoid, normalisation over the tokens).  The reference
        (tpmcl_utils.py:35-50) evaluates the same function pair by pair on repeat / repeat_interleave copies; fp32 throughout, the result
        differs from the pair-batch evaluation by summation order only."""
        ln, fc_a, act, fc_b, sig = self.attn_proj
        D = self.embed_dim
        qp = self.qk_proj(q_items.float().transpose(-2, -1)).transpose(-1, -2)          # [n_q, T, D]
        kf = k_items.float()                                                            # [n_k, T, D]
        N = float(qp.shape[1] * 2 * D)
        s1 = kf.sum(dim=(1, 2))[:, None] + qp.sum(dim=(1, 2))[None, :]                   # [n_k, n_q]
        s2 = (kf * kf).sum(dim=(1, 2))[:, None] + (qp * qp).sum(dim=(1, 2))[None, :]
        mu = s1 / N
        r = torch.rsqrt((s2 / N - mu * mu).clamp_min(0.0) + ln.eps)
        gam, bet, W = ln.weight.float(), ln.bias.float(), fc_a.weight.float()            # [T, 2D], [T, 2D], [D/2, 2D]
        A = torch.matmul(qp * gam[None, :, :D], W[:, :D].t())                            # [n_q, T, D/2]
        Bm = torch.matmul(kf * gam[None, :, D:], W[:, D:].t())                           # [n_k, T, D/2]
        c = torch.matmul(gam, W.t())                                                     # [T, D/2]
        d = torch.matmul(bet, W.t())
        h = r[:, :, None, None] * (Bm[:, None] + A[None, :]) - (r * mu)[:, :, None, None] * c + d     # [n_k, n_q, T, D/2]
        w = sig(fc_b(act(h))).squeeze(-1)                                                # [n_k, n_q, T]
        return w / w.sum(dim=-1, keepdim=True)


class TokenImportanceSelector(nn.Module):
    """Zero the most important tokens: those whose cumulative weight (descending order) is still below `thresh`
    (reference: tpmcl_utils.py:101-121).  Returns (masked tokens, keep policy)."""

    def __init__(self, thresh):
        super().__init__()
        self.register_buffer("thresh", thresh * torch.ones(1))

    def forward(self, x, attn_weight):
        w_sorted, order = attn_weight.sort(dim=1, descending=True)
        drop = torch.zeros_like(attn_weight).scatter(1, order, (w_sorted.cumsum(dim=1) < self.thresh).to(attn_weight.dtype))
        keep = 1.0 - drop
        return x * keep.unsqueeze(-1).to(x.dtype), keep


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
            self._run_init_tpmcl()
        if "wti" in self.interaction:
            def weight_head():  # l3_wti_arch 1: Linear(D, 1); 2 / 3: one / two hidden Linear(D, D) + ReLU in front (reference :35-53)
                layers = []
                for _ in range(int(self.wti_arch) - 1):
                    layers += [nn.Linear(hidden_size, hidden_size), nn.ReLU(inplace=True)]
                layers.append(nn.Linear(hidden_size, 1))
                return layers[0] if len(layers) == 1 else nn.Sequential(*layers)
            assert self.wti_arch in (1, 2, 3)
            self.text_weight_fc, self.video_weight_fc = weight_head(), weight_head()
        if self.sim_header == "seqTransf":
            self.frame_position_embeddings = nn.Embedding(77, hidden_size)
            self.transformerClip = TransformerClip(width=hidden_size, layers=self.cross_num_hidden_layers, heads=hidden_size // 64)

    def _run_init_tpmcl(self):
        embed_dim, max_frames = self.config.hidden_size, self.max_frames + 1   # (+1: the [SEP] token appended to the clip tokens)
        self.xwp_type = "linear"
        self.t2v_linear_xwp = LinearXWeightPredictor(num_frames=1, num_tokens=max_frames, embed_dim=embed_dim)
        self.v2t_linear_xwp = LinearXWeightPredictor(num_frames=max_frames, num_tokens=self.max_words, embed_dim=embed_dim)
        self.tis_selector = TokenImportanceSelector(self.config.get("l3_cis_thresh", 0.6))
        self.margin = float(self.config.get("l3_margin_loss_thresh", 0.6))

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
        x = feat.float()
        layers = list(fc) if isinstance(fc, nn.Sequential) else [fc]
        for m in layers[:-1]:   # wti_arch 2 / 3: hidden D x D layers on the fused GEMM; ReLU folded into its epilogue
            if isinstance(m, nn.Linear):
                x = HF.linear(x.to(torch.bfloat16).contiguous(), m.weight, m.bias, act="relu").float()
        last = layers[-1]
        z = torch.nn.functional.linear(x, last.weight.float(), last.bias.float()).squeeze(2)   # Linear(D, 1): a [*, D] x [D] reduction
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
        cap_output = (text_embed_l1.float().unsqueeze(1), cap_embed)
        simi = self._loose_similarity(cap_output, visual_embed, cap_mask, visual_mask, sim_header=self.sim_header)
        margin_loss = 0.0
        if self.training and self.partial_type > 0:
            margin_loss = self.get_partial_similarity(cap_output, visual_embed, cap_mask, visual_mask, self.partial_type)
        return simi, margin_loss

    # ------------------------------------------------------------------ TPM-CL (reference :280-523)
    def wti_interaction_row(self, text_feat, video_feat, text_mask, video_mask, nblocks=1):
        """One score per ALIGNED pair c (text_feat[c] vs video_feat[c]).  As in the reference, the token weights are contracted with
        'ct,bt->c' / 'cv,bv->c': the per-token maxima of every pair are weighted by the softmax weights SUMMED over the pair batch
        -- the pair batch being one 8 x 16 caption x video block: with `nblocks` > 1 the leading dimension holds that many blocks back
        to back and the sums stay inside each block."""
        text_feat, video_feat = text_feat.float(), video_feat.float()
        text_mask, video_mask = text_mask.float(), video_mask.float()
        if video_mask.shape[1] > video_feat.shape[1]:
            video_mask = video_mask[:, :1]
        elif video_mask.shape[1] != video_feat.shape[1]:
            video_mask = video_mask.repeat_interleave(video_feat.shape[1] // video_mask.shape[1], dim=1)
        if text_mask.shape[1] != text_feat.shape[1]:
            text_mask = text_mask[:, :1]
        logits = torch.einsum("ctd,cvd->ctv", text_feat, video_feat) * text_mask[:, :, None] * video_mask[:, None, :]
        t2v, v2t = logits.max(dim=-1).values, logits.max(dim=-2).values
        if "wti" in self.interaction:
            tw = self._masked_softmax(self.text_weight_fc, text_feat, text_mask)
            vw = self._masked_softmax(self.video_weight_fc, video_feat, video_mask)
            if nblocks > 1:
                per = tw.shape[0] // nblocks
                tws = tw.view(nblocks, per, -1).sum(1).repeat_interleave(per, 0)
                vws = vw.view(nblocks, per, -1).sum(1).repeat_interleave(per, 0)
                return ((t2v * tws).sum(1) + (v2t * vws).sum(1)) / 2.0
            return ((t2v * tw.sum(0)).sum(1) + (v2t * vw.sum(0)).sum(1)) / 2.0
        return (t2v.sum(1) / text_mask.sum(-1) + v2t.sum(1) / video_mask.sum(-1)) / 2.0

    def _loose_similarity_row(self, sequence_output, visual_output, attention_mask, video_mask, sim_header="meanP", nblocks=1):
        if "ti" not in self.interaction:
            raise NotImplementedError(f"interaction:{self.interaction} not implemented")
        return self.wti_interaction_row(sequence_output.contiguous(), visual_output.contiguous(), attention_mask, video_mask, nblocks)

    def _get_partial_output(self, sequence_output, visual_output, attention_mask, video_mask, xwp_type="linear", partial_type=-1):
        """The five [bt, bv] score matrices of one caption x video block that the margin losses use (full vs importance-masked tokens,
        sentence vs predicted global text feature).  Pair flattenings: "_i" = caption-major (p = i bv + j), "_j" = video-major
        (p = j bt + i), which is what repeat_interleave / repeat produce in the reference."""
        if xwp_type != "linear":
            raise NotImplementedError("attention predictor (the reference passes xwp_type='linear')")
        sent, words = sequence_output
        bt, bv = sent.shape[0], visual_output.shape[0]
        sent_j, wmask_j = sent.repeat(bv, 1, 1), attention_mask.repeat(bv, 1)
        words_i, wmask_i = words.repeat_interleave(bv, 0), attention_mask.repeat_interleave(bv, 0)
        vis_i, vmask_i = visual_output.repeat(bt, 1, 1), video_mask.repeat(bt, 1)
        vis_j, vmask_j = visual_output.repeat_interleave(bt, 0), video_mask.repeat_interleave(bt, 0)
        word_w = self.v2t_linear_xwp(vis_i, words_i)            # word importance given the video   [bt*bv, Nw]
        frame_w = self.t2v_linear_xwp(sent_j, vis_j)            # frame importance given the caption [bt*bv, V]
        glob = torch.einsum("abd,ab->ad", words_i.float(), word_w)
        glob = (glob / glob.norm(dim=-1, keepdim=True)).unsqueeze(1)
        out = dict.fromkeys(("t2vh", "t2vhh", "tg2vh", "tg2vhh", "tgh2vh"))
        if self.training and partial_type >= 2:
            words_masked, _ = self.tis_selector(words_i.float(), word_w)
            glob_partial = torch.einsum("abd,ab->ad", words_masked, word_w).unsqueeze(1)
            vis_masked, _ = self.tis_selector(vis_j.float(), frame_w)
            vis_partial, vmask_p, _ = self._agg_visual_feat(vis_masked, vmask_j, sim_header=self.sim_header)
            row = lambda a, b, ma, mb: self._loose_similarity_row(a, b, ma, mb, sim_header=self.sim_header)  # noqa: E731
            out["t2vhh"] = row(sent_j, vis_partial, wmask_j, vmask_p).reshape(bv, bt).t()
            out["t2vh"] = row(sent_j, vis_i, wmask_j, vmask_i).reshape(bt, bv)
            out["tg2vh"] = row(glob, vis_i, wmask_i, vmask_i).reshape(bt, bv)
            out["tg2vhh"] = row(glob, vis_partial, wmask_i, vmask_p).reshape(bv, bt).t()
            out["tgh2vh"] = row(glob_partial, vis_i, wmask_i, vmask_i).reshape(bt, bv)
        return out

    def _get_partial_output_blocks(self, sequence_output, visual_output, attention_mask, video_mask, bt=8, bv=16, partial_type=4):
        """ALL 8 x 16 caption x video blocks of _get_partial_output in one batched pass (the reference walks them in a Python double loop to bound
        its memory; on 288 GB the B_t x B_v pairs of a step fit at once): every per-pair operator is evaluated over the concatenation of the
        blocks' pair batches -- pair orderings inside a block as in the reference ("_i" caption-major, "_j" video-major) -- and the one
        operator that couples the pairs of a block (the token-weight sums of wti_interaction_row) keeps its sums inside the blocks.
        Returns the five [B_t, B_v] matrices.  Requires B_t % bt == 0 and B_v % bv == 0 (the loop handles ragged edges)."""
        sent, words = sequence_output
        Bt, Bv = sent.shape[0], visual_output.shape[0]
        nbt, nbv = Bt // bt, Bv // bv
        nb, dev = nbt * nbv, sent.device
        # pair batches as broadcast views (backward = a reduction over the broadcast axes, no index_put): block n = a nbv + b covers captions
        # a bt + i and videos b bv + j; caption-major pairs [a, b, i, j], video-major pairs [a, b, j, i]
        def cap(x, video_major):
            x = x.reshape(nbt, 1, 1, bt, *x.shape[1:]) if video_major else x.reshape(nbt, 1, bt, 1, *x.shape[1:])
            full = (nbt, nbv, bv, bt) if video_major else (nbt, nbv, bt, bv)
            return x.expand(*full, *x.shape[4:]).reshape(nb * bt * bv, *x.shape[4:])

        def vid(x, video_major):
            x = x.reshape(1, nbv, bv, 1, *x.shape[1:]) if video_major else x.reshape(1, nbv, 1, bv, *x.shape[1:])
            full = (nbt, nbv, bv, bt) if video_major else (nbt, nbv, bt, bv)
            return x.expand(*full, *x.shape[4:]).reshape(nb * bt * bv, *x.shape[4:])

        sent_j, wmask_j = cap(sent, True), cap(attention_mask, True)
        words_i, wmask_i = cap(words, False), cap(attention_mask, False)
        vis_i, vmask_i = vid(visual_output, False), vid(video_mask, False)
        vis_j, vmask_j = vid(visual_output, True), vid(video_mask, True)
        if self.v2t_linear_xwp._qk_same_embed_dim and self.t2v_linear_xwp._qk_same_embed_dim and not self.config.get("l3_xwp_pair_batch", False):
            # token-importance weights of all B_t x B_v pairs from per-caption / per-video pieces (LinearXWeightPredictor.forward_all_pairs),
            # then laid out in the pair orders the rest of this function uses: caption-major [a, b, i, j] / video-major [a, b, j, i]
            ww = self.v2t_linear_xwp.forward_all_pairs(visual_output, words)             # [B_t, B_v, Nw]   (k = words of caption i, q = video j)
            fw = self.t2v_linear_xwp.forward_all_pairs(sent, visual_output)              # [B_v, B_t, V]    (k = frames of video j, q = caption i)
            word_w = ww.reshape(nbt, bt, nbv, bv, -1).permute(0, 2, 1, 3, 4).reshape(nb * bt * bv, -1)
            frame_w = fw.reshape(nbv, bv, nbt, bt, -1).permute(2, 0, 1, 3, 4).reshape(nb * bt * bv, -1)
        else:
            word_w = self.v2t_linear_xwp(vis_i, words_i)
            frame_w = self.t2v_linear_xwp(sent_j, vis_j)
        glob = torch.einsum("abd,ab->ad", words_i.float(), word_w)
        glob = (glob / glob.norm(dim=-1, keepdim=True)).unsqueeze(1)
        words_masked, _ = self.tis_selector(words_i.float(), word_w)
        glob_partial = torch.einsum("abd,ab->ad", words_masked, word_w).unsqueeze(1)
        vis_masked, _ = self.tis_selector(vis_j.float(), frame_w)
        vis_partial, vmask_p, _ = self._agg_visual_feat(vis_masked, vmask_j, sim_header=self.sim_header)
        row = lambda x, y, mx, my: self._loose_similarity_row(x, y, mx, my, sim_header=self.sim_header, nblocks=nb)  # noqa: E731

        def grid(x, video_major):   # [nb * bt * bv] scores -> [B_t, B_v]
            blk = x.view(nb, bv, bt).transpose(1, 2) if video_major else x.view(nb, bt, bv)
            return blk.reshape(nbt, nbv, bt, bv).permute(0, 2, 1, 3).reshape(Bt, Bv)

        return {"t2vhh": grid(row(sent_j, vis_partial, wmask_j, vmask_p), True), "t2vh": grid(row(sent_j, vis_i, wmask_j, vmask_i), False),
                "tg2vh": grid(row(glob, vis_i, wmask_i, vmask_i), False), "tg2vhh": grid(row(glob, vis_partial, wmask_i, vmask_p), True),
                "tgh2vh": grid(row(glob_partial, vis_i, wmask_i, vmask_i), False)}

    def _get_partial_loss(self, sim_matrix, sim_matrix_bar):
        """MarginRankingLoss(margin)(diag(anchor), diag(partial), +1): the full-token score of a true pair must beat its
        importance-masked score by the margin."""
        if sim_matrix.shape[1] != sim_matrix_bar.shape[1]:
            sim_matrix_bar = sim_matrix_bar.repeat(1, sim_matrix.shape[1] // sim_matrix_bar.shape[1])
        return torch.clamp(self.margin - (torch.diagonal(sim_matrix) - torch.diagonal(sim_matrix_bar)), min=0).mean()

    def get_partial_similarity(self, sequence_output, visual_output, attention_mask, video_mask, partial_type=1):
        sent, words = sequence_output
        if not (self.training and partial_type >= 2):
            return 0.0
        names = ("t2vh", "t2vhh", "tg2vh", "tg2vhh", "tgh2vh")
        if sent.shape[0] % 8 == 0 and visual_output.shape[0] % 16 == 0 and sent.shape[0] * visual_output.shape[0] > 128 and not self.config.get("l3_partial_loop", False):
            M = self._get_partial_output_blocks(sequence_output, visual_output, attention_mask, video_mask, 8, 16, partial_type)
            return self._partial_losses(M, partial_type)
        rows = {n: [] for n in names}
        for t0 in range(0, sent.shape[0], 8):          # the reference's block sizes: 8 captions x 16 videos
            cols = {n: [] for n in names}
            blk = (sent[t0:t0 + 8], words[t0:t0 + 8])
            for v0 in range(0, visual_output.shape[0], 16):
                o = self._get_partial_output(blk, visual_output[v0:v0 + 16], attention_mask[t0:t0 + 8], video_mask[v0:v0 + 16],
                                             xwp_type="linear", partial_type=partial_type)
                for n in names:
                    cols[n].append(o[n])
            for n in names:
                rows[n].append(torch.cat(cols[n], dim=-1))
        M = {n: torch.cat(rows[n], dim=0) for n in names}
        return self._partial_losses(M, partial_type)

    def _partial_losses(self, M, partial_type):
        if get_world_size() > 1:
            M = {n: gather_tensor(m.contiguous(), method="cat", back_gradient=True, pad_tensors=True) for n, m in M.items()}
        loss = 0.0
        if partial_type in (2, 4):
            loss = loss + self._get_partial_loss(M["t2vh"], M["t2vhh"]) + self._get_partial_loss(M["tg2vh"], M["tg2vhh"])
        if partial_type in (3, 4):
            loss = loss + self._get_partial_loss(M["tg2vh"], M["tgh2vh"])
        return loss
