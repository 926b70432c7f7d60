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
