"""oracle.losses -- CPU restatement (plain torch fp32) of the contrastive losses on the hot path.
TEST INFRASTRUCTURE ONLY (see oracle/ops.py)."""
import torch


def mil_nce(sim, batch_size, n_pair=1, weight=None):
    """MIL-NCE over a [B*n, B*n] text-row x video-clip-column similarity matrix, closed form of
    prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:146-197:

        blk(k) = k // n
        l_r = LSE( S[:, r]  U  { S[r, c] : blk(c) != blk(r) } ) - LSE( { S[t, r] : blk(t) == blk(r) } )
        loss = mean_i  w_i * l_{i*n + n//2}

    There is no temperature; for n = 1 it is a (2B-1)-way softmax with target S[r, r]."""
    m = sim.shape[0]
    blk = torch.arange(m) // n_pair
    same = blk[:, None] == blk[None, :]  # [m, m]
    neg_inf = torch.finfo(sim.dtype).min
    col = sim.t()  # col[r, t] = S[t, r]
    row_other = sim.masked_fill(same, neg_inf)  # S[r, c] for c in other blocks
    denom = torch.logsumexp(torch.cat([col, row_other], dim=1), dim=1)
    numer = torch.logsumexp(col.masked_fill(~same, neg_inf), dim=1)
    per_row = denom - numer
    pick = torch.arange(batch_size) * n_pair + n_pair // 2
    chosen = per_row[pick]
    if weight is not None:
        chosen = chosen * weight
    return chosen.mean()


def clip_itc(img, txt, logit_scale):
    """Symmetric InfoNCE over logits = exp(logit_scale) * img @ txt^T.  The logits formula is the
    reference's (prj/M2_Encoder/m2_encoder.py:92-95; antmmf/modules/vision/backbone/clip/model.py:433-447);
    the reference ships no M2 training loss, so the symmetric cross-entropy is this
    build's statement of it ("parity unpinned" by the reference at the loss; pinned at the logits)."""
    logits = torch.exp(logit_scale) * img @ txt.t()
    tgt = torch.arange(img.shape[0])
    li = torch.logsumexp(logits, dim=1) - logits[tgt, tgt]
    lt = torch.logsumexp(logits, dim=0) - logits[tgt, tgt]
    return 0.5 * (li.mean() + lt.mean()), logits


def moco(pos, neg, temperature=0.05):
    """mean( LSE([pos, neg]/T) - LSE(pos/T) )  (prj/base_vtp/.../moco_utils.py:71-81)."""
    allv = torch.cat([pos, neg], dim=1) / temperature
    return (torch.logsumexp(allv, dim=1) - torch.logsumexp(pos / temperature, dim=1)).mean()


def cross_en(sim, logit_scale=100.0):
    """-mean diag log_softmax(scale * S)  (prj/dmae_vtp/.../dmae_utils.py:528-537)."""
    z = sim * logit_scale
    return (torch.logsumexp(z, dim=-1) - torch.diagonal(z)).mean()


def neg_nce(sim, logit_scale=100.0, pos_w=1.0, neg_w=0.5, margin=0.0):
    """NegNCE (dmae_utils.py:539-563): p = clamp(softmax(scale*S), 1e-6, 1-1e-6);
    positives -log p_ii; negatives -log(1 - p_ij) over off-diagonal (i, j) that violate the margin
    against either diagonal: relu(m + S_ij - S_ii) + relu(m + S_ij - S_jj) > 0."""
    p = torch.softmax(sim * logit_scale, dim=-1).clamp(1e-6, 1 - 1e-6)
    d = torch.diagonal(sim)
    viol = torch.relu(margin + sim - d[:, None]) + torch.relu(margin + sim - d[None, :])
    eye = torch.eye(sim.shape[0], dtype=torch.bool)
    hard = (viol > 0) & ~eye
    loss = pos_w * (-torch.log(torch.diagonal(p))).mean()
    if hard.any():
        loss = loss + neg_w * (-torch.log(1 - p[hard])).mean()
    return loss


def dmae_wti_similarity(text_feat, video_feat, text_mask, video_mask, text_weight=None, video_weight=None, self_weight=False,
                        weighted=True):
    """DmaeUtils._get_wti_similarity (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:85-131), restated with gathers.
    text_feat [A, T, D], video_feat [B, V, D], masks [A, T] / [B, V] (1 = real token), weights [A, T] / [B, V].
      M[a,b,t,v] = <text_a,t , video_b,v> * tmask[a,t] * vmask[b,v]        (masked entries are 0, not -inf)
      t2v[a,b,t] = max_v M  (+ 0.5 * f2f[b,z1] * M[a,b,t,z2] when self_weight: z1 = argmax_v M[a,b,t,:],
                   z2 = argmax_v' F[b,z1,v'], f2f[b,z1] = max_v' F[b,z1,v'], F[b] = masked frame-frame similarities, zero diagonal)
      v2t[a,b,v] = max_t M
      weighted ("wti"): out = ( sum_t t2v * tmask * tw  +  sum_v v2t * vmask * vw ) / 2
      else ("ti"):      out = ( sum_t t2v * tmask / sum(tmask)  +  sum_v v2t * vmask / sum(vmask) ) / 2"""
    tmask, vmask = text_mask.float(), video_mask.float()
    M = torch.einsum("atd,bvd->abtv", text_feat, video_feat) * tmask[:, None, :, None] * vmask[None, :, None, :]
    t2v, z1 = M.max(dim=-1)
    v2t = M.max(dim=-2).values
    if self_weight:
        F = torch.einsum("btd,bvd->btv", video_feat, video_feat) * vmask[:, :, None] * vmask[:, None, :]
        F = F * (1.0 - torch.eye(F.shape[1]))[None]
        f2f, z2_of = F.max(dim=-1)                                  # [B, V]
        A, B, T = z1.shape
        bidx = torch.arange(B)[None, :, None].expand(A, B, T)
        z2 = z2_of[bidx, z1]                                        # [A, B, T]
        t2v = t2v + 0.5 * f2f[bidx, z1] * torch.gather(M, 3, z2.unsqueeze(-1)).squeeze(-1)
    if weighted:
        t2v_s = (t2v * (tmask * text_weight)[:, None, :]).sum(-1)
        v2t_s = (v2t * (vmask * video_weight)[None, :, :]).sum(-1)
    else:
        t2v_s = (t2v * (tmask / tmask.sum(-1, keepdim=True))[:, None, :]).sum(-1)
        v2t_s = (v2t * (vmask / vmask.sum(-1, keepdim=True))[None, :, :]).sum(-1)
    return (t2v_s + v2t_s) / 2.0


def dmae_wti_interaction(P, text_feat, word_feat, video_feat, word_mask, video_mask, interaction="wti", with_va=True):
    """DmaeUtils.wti_interaction (dmae_utils.py:133-184), single process, one token per frame, wti_arch 1.
    text_feat [A, 1, D] sentence embedding, word_feat [A, Nw, D], video_feat [B, V, D]; P holds text_weight_fc / video_weight_fc.
    Weights = masked softmax of a Linear(D, 1) over the tokens; "wti"/"ti": sentence-vs-frames only; "att_wti"/"att_ti":
    mean of sentence-vs-frames and words-vs-frames."""
    def masked_softmax(feat, w, b, mask):
        z = (feat @ w.t()).squeeze(-1) + b
        return torch.softmax(z.masked_fill(mask < 0.5, float("-inf")), dim=-1)

    text_mask = word_mask[:, :1].float() if word_mask.shape[1] != text_feat.shape[1] else word_mask.float()
    weighted = "wti" in interaction
    tw = ww = vw = None
    if weighted:
        tw = masked_softmax(text_feat, P["text_weight_fc.weight"], P["text_weight_fc.bias"], text_mask)
        ww = masked_softmax(word_feat, P["text_weight_fc.weight"], P["text_weight_fc.bias"], word_mask.float())
        vw = masked_softmax(video_feat, P["video_weight_fc.weight"], P["video_weight_fc.bias"], video_mask.float())
    out = dmae_wti_similarity(text_feat, video_feat, text_mask, video_mask, tw, vw, with_va, weighted)
    if interaction in ("att_ti", "att_wti"):
        out = (out + dmae_wti_similarity(word_feat, video_feat, word_mask, video_mask, ww, vw, with_va, weighted)) / 2.0
    return out


# ------------------------------------------------------------------------------ DMAE TPM-CL (partial-order margin losses)
def _tpm_predictor(Pp, q, k):
    """LinearXWeightPredictor.forward with qdim = kdim = embed_dim (prj/dmae_vtp/roi_univl/univl/model/tpmcl_utils.py:6-50):
    q [P, F, D] is mapped along its token axis F -> T by a bias-free Linear, concatenated with k [P, T, D], LayerNorm over the whole
    [T, 2D] slab, Linear(2D, D/2) -> GELU(erf) -> Linear(D/2, 1) -> sigmoid, then normalised to sum 1 over the T tokens."""
    qq = torch.matmul(q.transpose(-2, -1), Pp["qk_proj.weight"].t()).transpose(-1, -2)
    qk = torch.cat([qq, k], dim=-1)
    h = torch.nn.functional.layer_norm(qk, qk.shape[-2:], Pp["attn_proj.0.weight"], Pp["attn_proj.0.bias"], 1e-5)
    h = torch.nn.functional.gelu(torch.matmul(h, Pp["attn_proj.1.weight"].t()))
    w = torch.sigmoid(torch.matmul(h, Pp["attn_proj.3.weight"].t())).squeeze(-1)
    return w / w.sum(dim=1, keepdim=True)


def _tpm_drop_important(x, w, thresh):
    """TokenImportanceSelector (tpmcl_utils.py:101-121): zero the most important tokens -- those whose cumulative weight, in
    descending order, is still below `thresh` -- and keep the rest."""
    sw, order = w.sort(dim=1, descending=True)
    drop_sorted = (sw.cumsum(dim=1) < thresh).to(w.dtype)
    drop = torch.zeros_like(w).scatter(1, order, drop_sorted)
    return x * (1.0 - drop).unsqueeze(-1)


def _tpm_row_similarity(P, text, video, tmask, vmask, weighted):
    """DmaeUtils.wti_interaction_row (dmae_utils.py:476-523): one score per aligned pair c (text[c] vs video[c]).
    NOTE the reference contracts the token weights with 'ct,bt->c' / 'cv,bv->c': every pair's max-logits are weighted by the SUM over
    the whole pair batch of the softmax weights -- restated as written."""
    if vmask.shape[1] != video.shape[1] and vmask.shape[1] > video.shape[1]:
        vmask = vmask[:, :1]
    elif vmask.shape[1] != video.shape[1]:
        vmask = vmask.repeat_interleave(video.shape[1] // vmask.shape[1], dim=1)
    if tmask.shape[1] != text.shape[1]:
        tmask = tmask[:, :1]
    logits = torch.einsum("ctd,cvd->ctv", text, video) * tmask[:, :, None] * vmask[:, None, :]
    t2v, v2t = logits.max(dim=-1).values, logits.max(dim=-2).values
    if weighted:
        def w(fcw, fcb, feat, mask):
            z = (feat @ fcw.t()).squeeze(-1) + fcb
            return torch.softmax(z.masked_fill(mask < 0.5, float("-inf")), dim=-1)
        tw = w(P["text_weight_fc.weight"], P["text_weight_fc.bias"], text, tmask)
        vw = w(P["video_weight_fc.weight"], P["video_weight_fc.bias"], video, vmask)
        return ((t2v * tw.sum(0)[None]).sum(1) + (v2t * vw.sum(0)[None]).sum(1)) / 2.0
    return (t2v.sum(1) / tmask.sum(-1) + v2t.sum(1) / vmask.sum(-1)) / 2.0


def dmae_tpmcl_margin_loss(P, text, word, video, word_mask, video_mask, partial_type, cis_thresh=0.6, margin=0.6, interaction="wti"):
    """DmaeUtils.get_partial_similarity in training mode (dmae_utils.py:280-388 with _get_partial_output :390-463), single process,
    meanP header, linear predictors.  text [B, 1, D] sentence, word [B, Nw, D], video [B, V, D].
    Blocks of 8 captions x 16 videos; inside a block every (caption i, video j) pair is formed in two flattenings
    ("row": p = i * bv + j, "batch": p = j * bt + i, exactly as repeat_interleave / repeat produce them).  Five block matrices are
    concatenated and only their DIAGONALS enter MarginRankingLoss(margin)(anchor, partial, +1):
      type 2: (t2vh, t2vhh) + (tg2vh, tg2vhh);  type 3: (tg2vh, tgh2vh);  type 4: all three."""
    weighted = "wti" in interaction
    B = text.shape[0]
    names = ("t2vh", "t2vhh", "tg2vh", "tg2vhh", "tgh2vh")
    rows = {n: [] for n in names}
    for t0 in range(0, B, 8):
        txt, wrd, wmask = text[t0:t0 + 8], word[t0:t0 + 8], word_mask[t0:t0 + 8]
        bt = txt.shape[0]
        cols = {n: [] for n in names}
        for v0 in range(0, B, 16):
            vid, vmask = video[v0:v0 + 16], video_mask[v0:v0 + 16]
            bv = vid.shape[0]
            txt_b, wmask_b = txt.repeat(bv, 1, 1), wmask.repeat(bv, 1)                        # p = j * bt + i
            wrd_r, wmask_r = wrd.repeat_interleave(bv, 0), wmask.repeat_interleave(bv, 0)      # p = i * bv + j
            vid_b, vmask_b = vid.repeat(bt, 1, 1), vmask.repeat(bt, 1)                         # p = i * bv + j
            vid_r, vmask_r = vid.repeat_interleave(bt, 0), vmask.repeat_interleave(bt, 0)      # p = j * bt + i
            t_w = _tpm_predictor(towers_sub(P, "v2t_linear_xwp."), vid_b, wrd_r)                # word weights given the video
            v_w = _tpm_predictor(towers_sub(P, "t2v_linear_xwp."), txt_b, vid_r)                # frame weights given the caption
            seq_g = torch.einsum("abd,ab->ad", wrd_r, t_w)
            seq_g = (seq_g / seq_g.norm(dim=-1, keepdim=True)).unsqueeze(1)                    # [bt*bv, 1, D], p = i * bv + j
            seq_gp = torch.einsum("abd,ab->ad", _tpm_drop_important(wrd_r, t_w, cis_thresh), t_w).unsqueeze(1)
            vid_p = _tpm_drop_important(vid_r, v_w, cis_thresh)                                 # p = j * bt + i
            sim = lambda a, b, ma, mb: _tpm_row_similarity(P, a, b, ma, mb, weighted)          # noqa: E731
            cols["t2vhh"].append(sim(txt_b, vid_p, wmask_b, vmask_r).reshape(bv, bt).t())
            cols["t2vh"].append(sim(txt_b, vid_b, wmask_b, vmask_b).reshape(bt, bv))
            cols["tg2vh"].append(sim(seq_g, vid_b, wmask_r, vmask_b).reshape(bt, bv))
            cols["tg2vhh"].append(sim(seq_g, vid_p, wmask_r, vmask_r).reshape(bv, bt).t())
            cols["tgh2vh"].append(sim(seq_gp, vid_b, wmask_r, vmask_b).reshape(bt, bv))
        for n in names:
            rows[n].append(torch.cat(cols[n], dim=-1))
    M = {n: torch.cat(rows[n], dim=0) for n in names}

    def rank_loss(anchor, partial):
        return torch.clamp(margin - (torch.diagonal(anchor) - torch.diagonal(partial)), min=0).mean()

    loss = 0.0
    if partial_type in (2, 4):
        loss = loss + rank_loss(M["t2vh"], M["t2vhh"]) + rank_loss(M["tg2vh"], M["tg2vhh"])
    if partial_type in (3, 4):
        loss = loss + rank_loss(M["tg2vh"], M["tgh2vh"])
    return loss


def towers_sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}
