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
