"""Row-sharded global-negative contrastive losses over RCCL (one process per GPU).

Reference behaviour being replaced (prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:313-325,357-387 with
antmmf/utils/distributed_utils.py:92-189): every rank all-gathers both embedding sets with a list-of-tensors
all_gather (+ a size exchange and host sync per call), computes the FULL [B_g*n]^2 similarity and loss
redundantly, and back-propagates through W serial `dist.reduce` calls.

Here each rank owns the B rows of its own pairs:
    forward   one all_gather_into_tensor per embedding set (fp32 [B, D], packed, static shapes: no size exchange),
              two [B x B_g] similarity slabs from the MFMA GEMM in fp32, the fused loss kernel on its rows only;
    backward  slab gradients -> four small GEMMs -> one reduce_scatter_tensor(sum) per embedding set.
The loss VALUE returned on every rank is the global mean (all-reduced, for logging parity with the reference,
where each rank computes the same global scalar); the embedding gradients are multiplied by the world size so
that, exactly as in the reference ("local.grad == W x single-process grad"), the subsequent
data-parallel MEAN of parameter gradients yields the true gradient of the global-batch loss.

Similarities are evaluated to fp32 accuracy on the bf16 MFMA pipe by splitting each fp32 operand into
hi + lo bf16 parts (three GEMMs: hi*hi + hi*lo + lo*hi); the GEMM is 0.01 % of the step's flops.
"""
import torch
import torch.distributed as dist

from . import ops

BF = torch.bfloat16


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


# Test switches of this module are Python attributes, not environment variables (round 6: the product path reads no ANTMMF_* variable except the library
# path ANTMMF_HIP_LIB).  FORCE_COLLECTIVES: run the collectives through a ONE-rank process group as well (tests/test_e2e_gpu.py::test_rccl_entry_points_one_rank
# -- the real RCCL entry points on the one-GPU boxes); SKIP_BATCH_CHECK: no equal-batch guard in front of the gathers.
FORCE_COLLECTIVES = False
SKIP_BATCH_CHECK = False


def _single(w):
    """world size 1 skips the collectives -- unless a process group exists and FORCE_COLLECTIVES is set (see above)."""
    return w == 1 and not (dist.is_available() and dist.is_initialized() and FORCE_COLLECTIVES)


def _assert_equal_batch(n_rows, device, group):
    """The sharded losses and the MoCo queue assume the same number of rows on every rank (row0 = rank * B, all_gather_into_tensor /
    reduce_scatter_tensor take equal shares; the reference instead exchanges sizes and pads on every call,
    distributed_utils.py:131-160).  A tiny all-gather of the local count + a DEVICE-side assert (no host sync) turns a ragged
    batch into an error instead of a hang or mis-indexed diagonals (on this image's wheel a failed device assert aborts the process: measured,
    profiles/r6_assert_async_probe.txt; a wheel built without device asserts would not notice -- `pad_ragged_batches` is the supported way to
    run ragged batches and checks on the host).  contrastive.SKIP_BATCH_CHECK = True removes it."""
    if SKIP_BATCH_CHECK:
        return
    w, _ = _world(group)
    mine = torch.full((1,), int(n_rows), dtype=torch.int64, device=device)
    every = torch.empty(w, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(every, mine, group=group)
    torch._assert_async((every == mine).all(), "ragged per-rank batch: the row-sharded losses need the same number of pairs on every rank "
                                               "(drop the last partial batch or pad it)")


def _all_gather(t, group):
    w, _ = _world(group)
    if _single(w):
        return t
    _assert_equal_batch(t.shape[0], t.device, group)
    out = torch.empty((w * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def _reduce_scatter_sum(full, group):
    w, _ = _world(group)
    if _single(w):
        return full
    out = torch.empty((full.shape[0] // w,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full.contiguous(), op=dist.ReduceOp.SUM, group=group)
    return out


def _all_gather_packed(tensors, group):
    """ONE all-gather for several per-rank tensors that share their leading (pair) dimension: they are packed side by side into a
    [B, sum of widths] buffer, gathered, and split again (xGMI collectives are latency- and per-link-bound: one message instead of
    one per embedding set -- image + text + the two VL heads of the M2 step, text + clips of the video step)."""
    w, _ = _world(group)
    if _single(w):
        return list(tensors)
    B = tensors[0].shape[0]
    flat = [t.reshape(B, -1) for t in tensors]
    widths = [f.shape[1] for f in flat]
    every = _all_gather(torch.cat(flat, dim=1), group)
    outs, off = [], 0
    for t, wd in zip(tensors, widths):
        outs.append(every[:, off:off + wd].reshape((every.shape[0],) + tuple(t.shape[1:])).contiguous())
        off += wd
    return outs


def _reduce_scatter_packed(fulls, group):
    """ONE reduce-scatter(sum) for several [B_g, ...] gradients (the inverse of _all_gather_packed)."""
    w, _ = _world(group)
    if _single(w):
        return list(fulls)
    Bg = fulls[0].shape[0]
    flat = [f.reshape(Bg, -1) for f in fulls]
    widths = [f.shape[1] for f in flat]
    mine = _reduce_scatter_sum(torch.cat(flat, dim=1), group)
    outs, off = [], 0
    for f, wd in zip(fulls, widths):
        outs.append(mine[:, off:off + wd].reshape((mine.shape[0],) + tuple(f.shape[1:])).contiguous())
        off += wd
    return outs


def _all_reduce_sum(t, group):
    w, _ = _world(group)
    if not _single(w):
        dist.all_reduce(t, group=group)
    return t


_DEFAULT_MAX_ROWS = None


def set_max_rows_per_rank(n):
    """Per-rank batch size of the run (BaseTrainer sets it from `training_parameters.batch_size` when `training_parameters.pad_ragged_batches` is on):
    the sharded losses then accept FEWER rows on any rank (the last, partial batch of an epoch) -- see _Rows.  None restores the equal-batch contract."""
    global _DEFAULT_MAX_ROWS
    _DEFAULT_MAX_ROWS = None if n is None else int(n)


class _Rows:
    """Where this rank's rows sit in the gathered batch.

    Equal batches (max_rows None, the default: what a DistributedSampler hands out, also for a short last batch): row0 = rank * B, every gathered
    row is real, Bg = W * B; a one-element all-gather + device-side assert guards the assumption.
    Ragged batches (max_rows given: the per-rank batch size of the configuration): the reference pads every gathered tensor to the largest rank and trims
    afterwards, with a size exchange + host sync per call (antmmf/utils/distributed_utils.py:131-160).  Here every rank pads its embeddings with zero rows to
    the STATIC max_rows, gathers [W * max_rows, ...], and gathers the W row counts once (device tensor, no host sync): padded columns of a similarity slab
    get -inf (they drop out of every log-sum-exp and receive zero gradient), padded local rows get coefficient 0, the mean divides by the true global
    count.  Shapes never depend on another rank's batch."""

    def __init__(self, B, device, group, max_rows=None):
        self.world, self.rank = _world(group)
        self.B, self.group = B, group
        if max_rows is None:
            max_rows = _DEFAULT_MAX_ROWS
        self.ragged = max_rows is not None and not _single(self.world)
        if not self.ragged:
            self.Bp, self.row0 = B, self.rank * B
            self.Bg = self.world * B      # python int
            return
        self.Bp, self.row0 = int(max_rows), self.rank * int(max_rows)
        mine = torch.full((1,), B, dtype=torch.int64, device=device)
        counts = torch.empty(self.world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(counts, mine, group=group)
        # a rank with MORE rows than the static maximum cannot be padded.  It takes part in the count exchange first and EVERY rank reads the gathered counts on the host, so
        # all ranks raise the same error together instead of the others blocking in the next collective behind a rank that raised alone.  (One host sync per loss call, on
        # the ragged path only -- the equal-batch default has none.  Round 5 used torch._assert_async here; whether a device-side assert fires depends on how the wheel was
        # built (on this image it does -- as a queue abort that takes the whole process down, profiles/r6_assert_async_probe.txt -- stock ROCm builds compile it out): a
        # host-side ValueError that every rank raises is the same on every build)
        most = int(counts.max())
        if most > int(max_rows):
            raise ValueError(f"a rank holds {most} rows, more than max_rows = {max_rows} (contrastive.set_max_rows_per_rank); this rank ({self.rank}) holds {B}")
        self.counts = counts
        self.Bg = counts.sum().to(torch.float32)                        # device scalar: the true global batch
        ar = torch.arange(self.Bp, device=device)
        valid_cols = (ar[None, :] < counts[:, None]).reshape(-1)        # [W * Bp]
        self.col_bias = torch.zeros(self.world * self.Bp, dtype=torch.float32, device=device).masked_fill_(~valid_cols, float("-inf"))
        self.row_valid = ar < B                                         # [Bp]

    def pad(self, t):
        """[B, ...] -> [Bp, ...] (zero rows)"""
        if not self.ragged or t.shape[0] == self.Bp:
            return t
        return torch.cat([t, t.new_zeros((self.Bp - t.shape[0],) + tuple(t.shape[1:]))], 0)

    def mask_cols(self, slab, per=1):
        """-inf into the columns of padded rows of other ranks (per = columns per gathered row: the clips of a video)"""
        if not self.ragged:
            return slab
        bias = self.col_bias if per == 1 else self.col_bias.repeat_interleave(per)
        return (slab + bias[None, :]).contiguous()

    def mask_rows(self, rows):
        """loss terms of padded local rows (their target column is a padded one: +inf) -> 0"""
        return torch.where(self.row_valid, rows, torch.zeros_like(rows)) if self.ragged else rows

    def coef(self, scale):
        """per-row backward coefficient scale / Bg (0 on padded rows)"""
        if not self.ragged:
            return None
        return torch.where(self.row_valid, scale / self.Bg, torch.zeros((), device=self.Bg.device)).float().contiguous()

    def unpad(self, g):
        return g[:self.B] if self.ragged and g.shape[0] != self.B else g


def _pad2(t, r_mult, c_mult):
    r, c = t.shape
    pr, pc = (-r) % r_mult, (-c) % c_mult
    if pr or pc:
        t = torch.nn.functional.pad(t, (0, pc, 0, pr))
    return t.contiguous()


def matmul_f32(A, B, a_rmajor=False, b_rmajor=False):
    """fp32-accurate  out[i, j] = sum_r A[i, r] B[j, r]  (operands [I, R] / [J, R], or [R, I] / [R, J] when *_rmajor)
    on the bf16 MFMA GEMM via a hi/lo split.  Operands are zero-padded to the kernel's alignment (tiny tensors) by the split kernel itself (ops.split_hi_lo: one launch
    per operand; round 6 -- it was a cast, a cast back, a subtraction, a cast, a pad and a copy each)."""
    I = A.shape[1] if a_rmajor else A.shape[0]
    J = B.shape[1] if b_rmajor else B.shape[0]
    ok = lambda t: t.dtype == torch.float32 and t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1])   # noqa: E731
    A = A if ok(A) else A.float().contiguous()
    B = B if ok(B) else B.float().contiguous()
    ah, al = ops.split_hi_lo(A)
    bh, bl = ops.split_hi_lo(B)
    Ip = ah.shape[1] if a_rmajor else ah.shape[0]
    Jp = bh.shape[1] if b_rmajor else bh.shape[0]
    # a long reduction into a handful of output tiles (the weight gradient of a small Linear over ~1e5 rows: the TPM-CL predictors) is split over the rows -- as ONE
    # workgroup per GEMM it took 2.1 ms x 3 (6.3 ms of the dmae12 step)
    split = 1
    if a_rmajor and b_rmajor:
        R = ah.shape[0]
        tiles = ((Ip + 127) // 128) * ((Jp + 127) // 128)
        if tiles < 64 and R >= 4096:
            split = min(64, max(1, 256 // tiles), R // 1024)
    # the first product initialises `out` (no zero-fill launch) where the layout's kernels store without accumulating: the forward layouts; the token-major (wgrad) layout
    # accumulates by construction (row split, atomics) and starts from zeros
    init = not (a_rmajor and b_rmajor) and split == 1
    out = torch.empty(Ip, Jp, dtype=torch.float32, device=A.device) if init else torch.zeros(Ip, Jp, dtype=torch.float32, device=A.device)
    for n, (x, y) in enumerate(((ah, bh), (ah, bl), (al, bh))):
        ops.gemm(x, y, out=out, p_rmajor=a_rmajor, q_rmajor=b_rmajor, accumulate=not (init and n == 0), split_k=split)
    return out[:I, :J] if (Ip != I or Jp != J) else out


class _MilNceSharded(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, clips, n_clips, weight, group, max_rows):
        B, D = text.shape
        rows = _Rows(B, text.device, group, max_rows)
        text, clips = rows.pad(text.float()).contiguous(), rows.pad(clips.float().view(B, n_clips * D)).contiguous()
        Bp = rows.Bp
        T_all, V_all = _all_gather_packed([text, clips], group)   # one message: [Bp, (1 + n) D]
        V_all = V_all.view(-1, D)
        centre = clips.view(Bp, n_clips, D)[:, n_clips // 2].contiguous()
        Rm = rows.mask_cols(matmul_f32(text, V_all).contiguous(), n_clips)      # [Bp, W Bp n]   <text_i, clip_c>
        Cm = rows.mask_cols(matmul_f32(centre, T_all).contiguous())             # [Bp, W Bp]     <centre clip of video_i, text_t>
        loss_rows, denom = ops.milnce_fwd(Rm, Cm, n_clips, rows.row0)
        if rows.ragged:
            coef = rows.coef(torch.ones((), device=text.device))
            loss_rows = rows.mask_rows(loss_rows)
        else:
            coef = torch.full((B,), 1.0 / rows.Bg, dtype=torch.float32, device=text.device)
        if weight is not None:
            coef = coef * rows.pad(weight.float())
        loss = _all_reduce_sum((loss_rows * coef).sum(), group)
        ctx.save_for_backward(text, centre, T_all, V_all, Rm, Cm, denom, coef)
        ctx.meta = (n_clips, rows, D)
        return loss

    @staticmethod
    def backward(ctx, gout):
        text, centre, T_all, V_all, Rm, Cm, denom, coef = ctx.saved_tensors
        n, rows, D = ctx.meta
        row0, Bp = rows.row0, rows.Bp
        dR, dC = ops.milnce_bwd(Rm, Cm, denom, (coef * gout * rows.world).contiguous(), n, row0, out_dtype=torch.float32)
        dT_all = matmul_f32(dC, centre, a_rmajor=True, b_rmajor=True)   # [Bg, D]   dC^T centre
        dV_all = matmul_f32(dR, text, a_rmajor=True, b_rmajor=True)     # [Bg*n, D] dR^T text
        dT_all[row0:row0 + Bp] += matmul_f32(dR, V_all, b_rmajor=True)  # dR V_all
        dVc = matmul_f32(dC, T_all, b_rmajor=True)                      # dC T_all -> centre clips of the local videos
        dV_all.view(-1, n, D)[row0:row0 + Bp, n // 2] += dVc
        dT, dV = _reduce_scatter_packed([dT_all, dV_all.view(-1, n * D)], rows.group)
        return rows.unpad(dT), rows.unpad(dV).reshape(-1, D), None, None, None, None


def mil_nce_sharded(text_embed, clip_embed, n_clips=1, weight=None, group=None, max_rows=None):
    """MIL-NCE over the global batch (get_mil_nce_loss, univl_video_ret.py:146-197), rows sharded over ranks.
    text_embed [B, D], clip_embed [B*n_clips, D]: this rank's L2-normalised embeddings.  max_rows: see _Rows (ragged per-rank batches)."""
    return _MilNceSharded.apply(text_embed, clip_embed, n_clips, weight, group, max_rows)


def _itc_term_fwd(im, tx, I_all, T_all, lsp, rows):
    """one symmetric InfoNCE term on this rank's rows: (local loss sum, tensors for backward)"""
    ls = lsp.detach().float().reshape(1).contiguous()
    xi = rows.mask_cols(matmul_f32(im, T_all).contiguous())   # image rows vs all texts
    xt = rows.mask_cols(matmul_f32(tx, I_all).contiguous())   # text rows vs all images
    li, lse_i = ops.softmax_ce_fwd(xi, rows.row0, ls)
    lt, lse_t = ops.softmax_ce_fwd(xt, rows.row0, ls)
    total = rows.mask_rows(li).sum() + rows.mask_rows(lt).sum()
    return total * (0.5 / rows.Bg), [im, tx, I_all, T_all, xi, xt, lse_i, lse_t, ls]


def _itc_term_bwd(saved, gout, rows):
    im, tx, I_all, T_all, xi, xt, lse_i, lse_t, ls = saved
    row0, Bp = rows.row0, rows.Bp
    if rows.ragged:
        coef = rows.coef(gout * rows.world * 0.5)
    else:
        coef = (gout * rows.world * 0.5 / rows.Bg).reshape(1).expand(Bp).contiguous().float()
    dscale = torch.zeros(1, dtype=torch.float32, device=im.device)
    dxi = ops.softmax_ce_bwd(xi, lse_i, coef, row0, ls, dscale=dscale, out_dtype=torch.float32)
    dxt = ops.softmax_ce_bwd(xt, lse_t, coef, row0, ls, dscale=dscale, out_dtype=torch.float32)
    dI_all = matmul_f32(dxt, tx, a_rmajor=True, b_rmajor=True)
    dT_all = matmul_f32(dxi, im, a_rmajor=True, b_rmajor=True)
    dI_all[row0:row0 + Bp] += matmul_f32(dxi, T_all, b_rmajor=True)
    dT_all[row0:row0 + Bp] += matmul_f32(dxt, I_all, b_rmajor=True)
    return dI_all, dT_all, (dscale * ls.exp()).reshape(())  # d/d log_scale = d/d s * s


class _ClipItcSharded(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, log_scale, group, max_rows):
        rows = _Rows(img.shape[0], img.device, group, max_rows)
        img, txt = rows.pad(img.float()).contiguous(), rows.pad(txt.float()).contiguous()
        I_all, T_all = _all_gather_packed([img, txt], group)
        loss, saved = _itc_term_fwd(img, txt, I_all, T_all, log_scale, rows)
        loss = _all_reduce_sum(loss, group)
        ctx.save_for_backward(*saved)
        ctx.rows = rows
        return loss

    @staticmethod
    def backward(ctx, gout):
        rows = ctx.rows
        dI_all, dT_all, dls = _itc_term_bwd(ctx.saved_tensors, gout, rows)
        dI, dT = _reduce_scatter_packed([dI_all, dT_all], rows.group)
        return rows.unpad(dI), rows.unpad(dT), dls, None, None


def clip_itc_sharded(img_embed, txt_embed, log_scale, group=None, max_rows=None):
    """Symmetric InfoNCE over logits = exp(log_scale) * img @ txt^T (prj/M2_Encoder/m2_encoder.py:92-95;
    antmmf/modules/vision/backbone/clip/model.py:442-444), rows sharded over ranks.  max_rows: see _Rows (ragged per-rank batches)."""
    return _ClipItcSharded.apply(img_embed, txt_embed, log_scale, group, max_rows)


class _ClipItcPairSharded(torch.autograd.Function):
    """The M2 step's TWO symmetric InfoNCE terms (cls heads with logit_scale, VL-FFN heads with logit_vl_scale) with ONE
    exchange each way: the four [B, D] embedding sets travel in a single all-gather ([B, 4 D]), the two loss values in one all-reduce,
    the four embedding gradients in a single reduce-scatter.  Arithmetic per term is _ClipItcSharded's."""

    @staticmethod
    def forward(ctx, img1, txt1, ls1, img2, txt2, ls2, group, max_rows):
        rows = _Rows(img1.shape[0], img1.device, group, max_rows)
        locs = [rows.pad(t.float()).contiguous() for t in (img1, txt1, img2, txt2)]
        alls = _all_gather_packed(locs, group)
        saved, losses = [], []
        for (im, tx, I_all, T_all, lsp) in ((locs[0], locs[1], alls[0], alls[1], ls1), (locs[2], locs[3], alls[2], alls[3], ls2)):
            l, sv = _itc_term_fwd(im, tx, I_all, T_all, lsp, rows)
            losses.append(l)
            saved += sv
        loss = _all_reduce_sum(torch.stack(losses), group)
        ctx.save_for_backward(*saved)
        ctx.rows = rows
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g1, g2):
        rows = ctx.rows
        sv = ctx.saved_tensors
        grads, dlss = [], []
        for k, gout in enumerate((g1, g2)):
            dI_all, dT_all, dls = _itc_term_bwd(sv[9 * k:9 * k + 9], gout, rows)
            grads += [dI_all, dT_all]
            dlss.append(dls)
        dI1, dT1, dI2, dT2 = [rows.unpad(g) for g in _reduce_scatter_packed(grads, rows.group)]
        return dI1, dT1, dlss[0], dI2, dT2, dlss[1], None, None


def clip_itc_pair_sharded(img1, txt1, log_scale1, img2, txt2, log_scale2, group=None, max_rows=None):
    """(loss of pair 1, loss of pair 2) -- two clip_itc_sharded terms with packed collectives (one all-gather, one loss all-reduce,
    one reduce-scatter for both).  max_rows: see _Rows (ragged per-rank batches)."""
    return _ClipItcPairSharded.apply(img1, txt1, log_scale1, img2, txt2, log_scale2, group, max_rows)


# ------------------------------------------------------------------------------ MoCo (queue negatives)
class _MocoLoss(torch.autograd.Function):
    """mean_i [ LSE({<q_i, kpos_i,c>}_c U {<q_i, queue_:,k>}_k) / T - LSE({<q_i, kpos_i,c>}_c / T) ]
    (MocoUtils.moco_loss + the einsums around it, prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:288-311,
    moco_utils.py:71-81).  q [R, D] carries the gradient; kpos [R, Np, D] and queue [D, K] are keys (no gradient).
    The [R, K] negatives come from the fp32-accurate split GEMM and are consumed by one fused row kernel."""

    @staticmethod
    def forward(ctx, q, kpos, queue, temperature):
        qf = q.float().contiguous()
        pos = (qf.unsqueeze(1) * kpos.float()).sum(-1).contiguous()       # [R, Np]  (R * Np * D multiply-adds: tiny)
        neg = matmul_f32(qf, queue, b_rmajor=True)                         # [R, K]
        loss_rows, lse_all, lse_pos = ops.moco_fwd(pos, neg, temperature)
        # the queue is overwritten in place by dequeue_and_enqueue before backward runs: keep the bf16 operand the dq GEMM needs
        qb = _pad2(queue.to(BF), 8, 8)                                     # [D, K]: r-contiguous Q operand (j = D, r = K)
        ctx.save_for_backward(pos, neg, lse_all, lse_pos, kpos, qb)
        ctx.temperature, ctx.q_dtype, ctx.dim = float(temperature), q.dtype, queue.shape[0]
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, g):
        pos, neg, lse_all, lse_pos, kpos, qb = ctx.saved_tensors
        R = pos.shape[0]
        coef = (g.float() / R).expand(R).contiguous()
        dpos, dneg = ops.moco_bwd(pos, neg, lse_all, lse_pos, coef, ctx.temperature, out_dtype=BF)
        dq = ops.gemm(_pad2(dneg, 8, 8), qb, out_dtype=torch.float32)[:R, :ctx.dim]
        dq = dq + (dpos.unsqueeze(-1) * kpos.float()).sum(1)
        return dq.to(ctx.q_dtype), None, None, None


def moco_loss(q, kpos, queue, temperature):
    """q [R, D] (grad), kpos [R, Np, D] positive keys per row, queue [D, K] negative keys."""
    return _MocoLoss.apply(q, kpos.detach(), queue.detach(), temperature)


# ------------------------------------------------------------------------------ DMAE losses on a square similarity matrix
class _NegNCE(torch.autograd.Function):
    """NegNCE (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:539-563): c_pos * mean_i(-log p_ii) + c_neg * mean over the
    margin-violating negatives of -log(1 - p_ij), p = clamp(softmax(scale * S)).  Fused row kernels; the count of selected
    negatives stays on the device (no host sync)."""

    @staticmethod
    def forward(ctx, S, scale, c_pos, c_neg, margin):
        Sf = S.float().contiguous()
        diag = torch.diagonal(Sf).contiguous()
        pos, nsum, ncnt, lse = ops.negnce_fwd(Sf, diag, 0, scale, margin)
        cnt = ncnt.sum()
        loss = c_pos * pos.mean() + torch.where(cnt > 0, c_neg * nsum.sum() / cnt.clamp_min(1.0), torch.zeros_like(cnt))
        ctx.save_for_backward(Sf, diag, lse, cnt)
        ctx.cfg = (float(scale), float(c_pos), float(c_neg), float(margin), S.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        Sf, diag, lse, cnt = ctx.saved_tensors
        scale, c_pos, c_neg, margin, dt = ctx.cfg
        coef = torch.stack([g.float() * c_pos / Sf.shape[0], torch.where(cnt > 0, g.float() * c_neg / cnt.clamp_min(1.0), torch.zeros_like(cnt))]).contiguous()
        dS = ops.negnce_bwd(Sf, diag, lse, coef, 0, scale, margin, out_dtype=torch.float32)
        return dS.to(dt), None, None, None, None


def neg_nce(sim_matrix, logit_scale=100.0, c_pos_w=1.0, c_neg_w=0.5, margin=0.0):
    return _NegNCE.apply(sim_matrix, logit_scale, c_pos_w, c_neg_w, margin)


class _CrossEn(torch.autograd.Function):
    """CrossEn (dmae_utils.py:528-537): -mean_i log_softmax(scale * S)_ii, on the fused softmax-CE row kernels."""

    @staticmethod
    def forward(ctx, S, scale):
        Sf = S.float().contiguous()
        rows, lse = ops.softmax_ce_fwd(Sf, 0, None, scale)
        ctx.save_for_backward(Sf, lse)
        ctx.cfg = (float(scale), S.dtype)
        return rows.mean()

    @staticmethod
    def backward(ctx, g):
        Sf, lse = ctx.saved_tensors
        scale, dt = ctx.cfg
        coef = (g.float() / Sf.shape[0]).expand(Sf.shape[0]).contiguous()
        return ops.softmax_ce_bwd(Sf, lse, coef, 0, None, scale, None, out_dtype=torch.float32).to(dt), None


def cross_en(sim_matrix, logit_scale=100.0):
    return _CrossEn.apply(sim_matrix, logit_scale)


# ------------------------------------------------------------------------------ MIL-NCE on an explicit [T, V] matrix (stage 2)
class _MilNceMatrix(torch.autograd.Function):
    """get_mil_nce_loss (univl_video_ret.py:146-197) for n_pair = 1 on a given square similarity matrix S[t, v] (the stage-2
    cross-encoder scores, :389-443): the fused row kernels with Rm = S and Cm = S^T; optional per-row weights (:192-195)."""

    @staticmethod
    def forward(ctx, S, weight):
        Sf = S.float().contiguous()
        St = Sf.t().contiguous()
        rows, denom = ops.milnce_fwd(Sf, St, 1, 0)
        ctx.save_for_backward(Sf, St, denom, weight)
        ctx.dt = S.dtype
        return (rows * weight).mean() if weight is not None else rows.mean()

    @staticmethod
    def backward(ctx, g):
        Sf, St, denom, weight = ctx.saved_tensors
        B = Sf.shape[0]
        coef = (g.float() / B).expand(B)
        if weight is not None:
            coef = coef * weight.float()
        dR, dC = ops.milnce_bwd(Sf, St, denom, coef.contiguous(), 1, 0, out_dtype=torch.float32)
        return (dR + dC.t()).to(ctx.dt), None


def mil_nce_matrix(S, weight=None):
    return _MilNceMatrix.apply(S, weight)


# ------------------------------------------------------------------------------ DMAE token-wise interaction
class _WtiReduce(torch.autograd.Function):
    """(t2v [A,B,T], v2t [A,B,V]) of DmaeUtils._get_wti_similarity for one block of text rows: the fp32-accurate split GEMM gives
    the [A*T, B*V] similarity slab, one fused kernel reduces it (max over frames / words, optional second-best-frame term) and
    records the arg-max routing; backward rebuilds the sparse slab gradient in one pass and returns to the features with two GEMMs."""

    @staticmethod
    def forward(ctx, text, video, tmask, vmask, f2f, z2_of):
        A, T, D = text.shape
        B, V, _ = video.shape
        t2d, v2d = text.reshape(A * T, D).float().contiguous(), video.reshape(B * V, D).float().contiguous()
        S = matmul_f32(t2d, v2d).contiguous()                               # [A*T, B*V] (matmul_f32 slices its padded result)
        tmask, vmask = tmask.float().contiguous(), vmask.float().contiguous()
        f2f_c = None if f2f is None else f2f.detach().float().contiguous()
        z2_c = None if z2_of is None else z2_of.to(torch.int32).contiguous()
        t2v, v2t, z1, tmax = ops.wti_reduce_fwd(S, A, T, B, V, tmask, vmask, f2f_c, z2_c)
        ctx.save_for_backward(S, t2d, v2d, tmask, vmask, f2f_c, z2_c, z1, tmax)
        ctx.dims = (A, T, B, V, D, text.dtype, video.dtype)
        return t2v, v2t

    @staticmethod
    def backward(ctx, dt2v, dv2t):
        S, t2d, v2d, tmask, vmask, f2f_c, z2_c, z1, tmax = ctx.saved_tensors
        A, T, B, V, D, tdt, vdt = ctx.dims
        dS, df2f = ops.wti_reduce_bwd(S, A, T, B, V, tmask, vmask, f2f_c, z2_c, z1, tmax, dt2v.float(), dv2t.float(), out_dtype=BF)
        dSp = _pad2(dS, 8, 8)
        vb, tb = _pad2(v2d.to(BF), 8, 8), _pad2(t2d.to(BF), 8, 8)
        # dtext[i, d] = sum_j dS[i, j] video[j, d];  dvideo[j, d] = sum_i dS[i, j] text[i, d]
        dtext = ops.gemm(dSp, vb, q_rmajor=True, out_dtype=torch.float32)[:A * T, :D]
        dvideo = ops.gemm(dSp, tb, p_rmajor=True, q_rmajor=True, out_dtype=torch.float32)[:B * V, :D]
        return dtext.reshape(A, T, D).to(tdt), dvideo.reshape(B, V, D).to(vdt), None, None, df2f, None


class _FrameTable(torch.autograd.Function):
    """F[b] = X[b] X[b]^T for X [B, V, D] (the frame-frame cosines of each video, DmaeUtils._get_wti_similarity's second-best-frame term,
    dmae_utils.py:98-119) on the fp32-accurate split MFMA GEMM: videos are taken in groups, one [G V, G V] product per group whose diagonal
    V x V blocks are the answer (a batched [V, D] x [D, V] product is 12 x 12 outputs per video: no tile shape fits it).  Backward: dX = (dF + dF^T) X per
    video, as one product of the block-diagonal gradient with the group's rows."""

    GROUP_ROWS = 2048

    @staticmethod
    def _groups(B, V):
        g = max(1, _FrameTable.GROUP_ROWS // V)
        return [(b0, min(B, b0 + g)) for b0 in range(0, B, g)]

    @staticmethod
    def forward(ctx, X):
        B, V, D = X.shape
        Xf = X.float().contiguous()
        out = torch.empty(B, V, V, dtype=torch.float32, device=X.device)
        for b0, b1 in _FrameTable._groups(B, V):
            G = b1 - b0
            rows = Xf[b0:b1].reshape(G * V, D)
            S = matmul_f32(rows, rows).view(G, V, G, V)
            idx = torch.arange(G, device=X.device)
            out[b0:b1] = S[idx, :, idx, :]
        ctx.save_for_backward(Xf)
        ctx.dt = X.dtype
        return out

    @staticmethod
    def backward(ctx, dF):
        (Xf,) = ctx.saved_tensors
        B, V, D = Xf.shape
        dX = torch.empty_like(Xf)
        sym = (dF + dF.transpose(1, 2)).float()
        for b0, b1 in _FrameTable._groups(B, V):
            G = b1 - b0
            blk = torch.zeros(G, V, G, V, dtype=torch.float32, device=Xf.device)
            idx = torch.arange(G, device=Xf.device)
            blk[idx, :, idx, :] = sym[b0:b1]
            rows = Xf[b0:b1].reshape(G * V, D)
            dX[b0:b1] = matmul_f32(blk.view(G * V, G * V), rows, b_rmajor=True).view(G, V, D)
        return dX.to(ctx.dt)


def frame_table(video_feat):
    """[B, V, D] -> [B, V, V] per-video Gram matrices (see _FrameTable)."""
    return _FrameTable.apply(video_feat)


def wti_similarity(text_feat, video_feat, text_mask, video_mask, text_weight=None, video_weight=None, self_weight=False, weighted=True,
                   rows_per_block=None):
    """DmaeUtils._get_wti_similarity (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:85-131) -> [A, B].
    text_feat [A, T, D], video_feat [B, V, D] (L2-normalised by the caller), masks 1 = real token; see oracle.losses.dmae_wti_similarity
    for the arithmetic.  Text rows are processed in blocks so that the fp32 slab stays below ~1 GiB."""
    A, T, _ = text_feat.shape
    B, V, _ = video_feat.shape
    tmask, vmask = text_mask.float(), video_mask.float()
    f2f = z2_of = None
    if self_weight:  # per-video frame-frame table: [B, V, V], tiny next to the text-video slab
        F = frame_table(video_feat) * vmask[:, :, None] * vmask[:, None, :]
        F = F * (1.0 - torch.eye(V, device=F.device, dtype=F.dtype))[None]
        f2f, z2_of = F.max(dim=-1)
    if rows_per_block is None:
        rows_per_block = max(1, (1 << 28) // max(1, T * B * V))
    t2v_s, v2t_s = [], []
    for a0 in range(0, A, rows_per_block):
        t2v, v2t = _WtiReduce.apply(text_feat[a0:a0 + rows_per_block].contiguous(), video_feat, tmask[a0:a0 + rows_per_block], vmask, f2f, z2_of)
        tm = tmask[a0:a0 + rows_per_block]
        if weighted:
            t2v_s.append((t2v * (tm * text_weight[a0:a0 + rows_per_block].float())[:, None, :]).sum(-1))
            v2t_s.append((v2t * (vmask * video_weight.float())[None, :, :]).sum(-1))
        else:
            t2v_s.append((t2v * (tm / tm.sum(-1, keepdim=True))[:, None, :]).sum(-1))
            v2t_s.append((v2t * (vmask / vmask.sum(-1, keepdim=True))[None, :, :]).sum(-1))
    return (torch.cat(t2v_s, 0) + torch.cat(v2t_s, 0)) / 2.0
