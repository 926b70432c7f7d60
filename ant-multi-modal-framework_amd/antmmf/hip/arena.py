"""Flat parameter arena + fused AdamW (MI355X memory layout for the training step).

All parameters of a model live in ONE fp32 master buffer, with a same-shaped fp32 gradient buffer and a
bf16 compute shadow; `nn.Parameter.data` / `.grad` become views.  Consequences:
  * the wgrad GEMMs accumulate straight into the gradient arena (antmmf.hip.functional.GradSink);
  * data-parallel gradient reduction is a handful of large RCCL all-reduces over contiguous memory
    (sized for xGMI's per-link bandwidth) instead of DDP's per-bucket copies;
  * the optimizer is one kernel launch per parameter group that also rewrites the bf16 shadow.
Parameter groups keep the reference's semantics (lr / weight_decay per group:
prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:482-542); groups are laid out contiguously.
"""
import torch
import torch.distributed as dist

from . import ops

ALIGN = 64  # elements; keeps every view 16-B aligned in all three buffers


def _force_collectives():
    """tests only: antmmf.hip.contrastive.FORCE_COLLECTIVES (the collectives through a one-rank process group)"""
    from . import contrastive

    return contrastive.FORCE_COLLECTIVES


def tag_pack(*params):
    """Ask the arena to lay these parameters out ADJACENTLY, in this order (they must end up in one parameter group): the separate q / k / v projection weights (and
    biases) of an attention module, which the fused layer consumes as ONE [3d, d] GEMM operand -- adjacent in the arena, the packed operand is a VIEW of the bf16 shadow
    (functional._packed_qkv_weight) instead of three copies per layer and optimizer step.  Called by the parameter-holder modules at construction; harmless without an arena."""
    token = object()
    for i, p in enumerate(params):
        if p is not None:
            p._antmmf_pack = (token, i)


def _pack_order(ps):
    """`ps` with the members of every tagged pack pulled together (in tag order) at the position of the pack's first member; everything else keeps its place."""
    by_token = {}
    for p in ps:
        t = getattr(p, "_antmmf_pack", None)
        if t is not None:
            by_token.setdefault(id(t[0]), []).append(p)
    out, placed = [], set()
    for p in ps:
        if id(p) in placed:
            continue
        t = getattr(p, "_antmmf_pack", None)
        members = sorted(by_token[id(t[0])], key=lambda q: q._antmmf_pack[1]) if t is not None else [p]
        for q in members:
            placed.add(id(q))
            out.append(q)
    return out


class ParamArena:
    def __init__(self, param_groups, device=None):
        """param_groups: list of dicts with "params" (as torch optimizers take them)."""
        seen, self.groups = set(), []
        total = legacy_total = 0
        self.legacy_offsets = {}     # id(p) -> offset in the layout WITHOUT pack reordering (what builds before round 6 used: HipAdamW.load_state_dict remaps their flat moments)
        for g in param_groups:
            ps = []
            for p in g["params"]:
                if id(p) in seen or not p.requires_grad:
                    continue
                seen.add(id(p))
                ps.append(p)
            for p in ps:
                self.legacy_offsets[id(p)] = legacy_total
                legacy_total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            ps = _pack_order(ps)
            start = total
            for p in ps:
                total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.groups.append(dict(params=ps, start=start, end=total, opts={k: v for k, v in g.items() if k != "params"}))
        if total == 0:
            raise ValueError("ParamArena: no trainable parameters")
        device = device or self.groups[0]["params"][0].device
        self.master = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(total, dtype=torch.bfloat16, device=device)
        off = 0
        for g in self.groups:
            for p in g["params"]:
                n = p.numel()
                view = self.master[off:off + n].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p._antmmf_main_grad = self.grad[off:off + n].view(p.shape)
                p._antmmf_bf16 = self.shadow[off:off + n].view(p.shape)
                p._antmmf_arena, p._antmmf_offset = self, off   # lets a MoCo key tower mirror this layout (one-launch EMA)
                p.grad = p._antmmf_main_grad
                p.register_hook(lambda g, p=p: self._late_grad_guard(p, g))
                off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.sync_shadow()

    def sync_shadow(self):
        """Rebuild the whole bf16 shadow from the fp32 masters (one cast launch).  Required after writes that torch's version counter does
        not see (`p.data...` edits); ordinary in-place writes to a parameter are detected per parameter by functional.compute_copy."""
        ops.cast_bf16(self.master, out=self.shadow)
        for g in self.groups:
            for p in g["params"]:
                p._antmmf_ver = p._version
        from .functional import bump_weight_version

        bump_weight_version()

    def zero_grad(self):
        self.grad.zero_()

    def grad_norm(self):
        s = torch.zeros(1, dtype=torch.float32, device=self.grad.device)
        ops.sumsq_(s, self.grad)
        return s.sqrt()

    # ------------------------------------------------------------------ data-parallel gradient reduction
    def _build_buckets(self, bucket_bytes):
        """Contiguous ranges of whole parameters, >= bucket_bytes each (xGMI ring all-reduce is per-link-bound: few large messages)."""
        want = max(1, bucket_bytes // 4)
        buckets, start, params = [], 0, []
        for g in self.groups:
            off = g["start"]
            for p in g["params"]:
                n = (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                params.append(p)
                off += n
                if off - start >= want:
                    buckets.append(dict(start=start, end=off, params=params))
                    start, params = off, []
        if params or start < self.grad.numel():
            buckets.append(dict(start=start, end=self.grad.numel(), params=params))
        for i, b in enumerate(buckets):
            for p in b["params"]:
                p._antmmf_bucket = i
        return buckets

    def arm_overlap(self, group=None, bucket_bytes=256 << 20, reduce_dtype=None):
        """Call BEFORE the forward pass of an optimizer-step iteration: the fused layers then report (functional._TransformerLayer)
        when a parameter's gradient is final, and every bucket whose parameters are all final is all-reduced right away on RCCL's
        stream, under the rest of the backward pass.  `allreduce_grads()` launches what is left and waits.
        reduce_dtype=torch.bfloat16 sends buckets as bf16 (half the xGMI bytes; the sum of W bf16 values carries ~3 significant digits)."""
        self._ov = None
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _force_collectives()):
            return False
        key = (bucket_bytes,)
        if getattr(self, "_bucket_key", None) != key:
            self._buckets, self._bucket_key = self._build_buckets(bucket_bytes), key
        for g in self.groups:
            for p in g["params"]:
                p._antmmf_uses = 0
                p._antmmf_mixed = False
        self._ov = dict(group=group, dtype=reduce_dtype, handles=[], launched=set(), frozen=False, left=None)
        return True

    def _late_grad_guard(self, p, g):
        """torch's own AccumulateGrad is about to add `g` to p.grad (a path outside the fused layers: plain torch ops, a tied weight; the
        fused nodes hand autograd None for arena parameters, which arrives here as g = None and is not a gradient).  If the bucket of that
        parameter has already been handed to RCCL the contribution would be lost on the other ranks -> refuse loudly."""
        ov = getattr(self, "_ov", None)
        if g is not None and ov is not None and getattr(p, "_antmmf_bucket", None) in ov["launched"]:
            raise RuntimeError(f"antmmf.hip.arena: a gradient reached a parameter (shape {tuple(p.shape)}, arena offset {p._antmmf_offset}, uses left "
                               f"{getattr(p, '_antmmf_uses', None)}) after its bucket's all-reduce had started (a parameter used by a "
                               "fused layer AND by an untracked op); call arena.note_untracked([p]) in the forward pass or disable overlap_grad_allreduce")

    def note_untracked(self, params):
        """Forward-pass notice from an autograd node that writes these parameters' gradients without reporting back (Linear, LayerNorm,
        embeddings, patch embed): their buckets are reduced after the backward pass, whatever the fused layers report."""
        ov = getattr(self, "_ov", None)
        if ov is None or ov["frozen"]:
            return
        for p in params:
            if p is not None and getattr(p, "_antmmf_arena", None) is self:
                p._antmmf_mixed = True

    def note_forward(self, params):
        ov = getattr(self, "_ov", None)
        if ov is None or ov["frozen"]:
            return
        for p in params:
            if p is not None and getattr(p, "_antmmf_arena", None) is self:
                p._antmmf_uses += 1

    def note_backward(self, params):
        """The calling autograd node has accumulated its share of these parameters' gradients."""
        ov = getattr(self, "_ov", None)
        if ov is None:
            return
        if not ov["frozen"]:  # first backward node of the step: per bucket, how many tracked parameters are still open
            ov["frozen"] = True
            ov["left"] = []
            for b in self._buckets:
                tracked = sum(1 for p in b["params"] if p._antmmf_uses > 0)
                untracked = sum(1 for p in b["params"] if p._antmmf_uses == 0 or getattr(p, "_antmmf_mixed", False))
                ov["left"].append(tracked if untracked == 0 else -1)  # -1: holds parameters outside the fused layers -> reduced at the end
        for p in params:
            if p is None or getattr(p, "_antmmf_arena", None) is not self or p._antmmf_uses <= 0:
                continue
            p._antmmf_uses -= 1
            if p._antmmf_uses == 0:
                i = p._antmmf_bucket
                if ov["left"][i] > 0:
                    ov["left"][i] -= 1
                    if ov["left"][i] == 0:
                        self._launch_bucket(i)

    def _launch_bucket(self, i):
        ov, b = self._ov, self._buckets[i]
        if i in ov["launched"]:
            return
        ov["launched"].add(i)
        # launch order of the step, for diagnosing the first multi-GPU runs: (bucket, MiB, from inside backward?, host time) -- `last_bucket_log` after the step
        import time

        ov.setdefault("log", []).append((i, (b["end"] - b["start"]) * 4 >> 20, not ov.get("closing", False), time.perf_counter()))
        seg = self.grad[b["start"]:b["end"]]
        if ov["dtype"] is not None and ov["dtype"] != torch.float32:
            tmp = seg.to(ov["dtype"])
            ov["handles"].append((dist.all_reduce(tmp, group=ov["group"], async_op=True), tmp, seg))
        else:
            ov["handles"].append((dist.all_reduce(seg, group=ov["group"], async_op=True), None, None))

    def allreduce_grads(self, group=None, bucket_bytes=512 << 20, reduce_dtype=None):
        """SUM all-reduce of the gradient arena in large contiguous buckets (the 1/world average is folded into the optimizer's
        grad_scale).  After arm_overlap(): launches the buckets the backward pass has not sent yet and waits for all of them.
        Returns the world size."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(group)
        if world == 1 and not _force_collectives():
            return 1
        if getattr(self, "_ov", None) is None:
            self.arm_overlap(group, bucket_bytes, reduce_dtype)
            self._ov["frozen"], self._ov["left"] = True, [-1] * len(self._buckets)
        self._ov["closing"] = True
        import time

        t_close = time.perf_counter()
        for i in range(len(self._buckets)):
            self._launch_bucket(i)
        for h, tmp, seg in self._ov["handles"]:
            h.wait()
            if tmp is not None:
                seg.copy_(tmp)
        self.overlapped_buckets = len(self._ov["launched"]) - sum(1 for x in self._ov["left"] if x != 0)   # diagnostics / tests
        # per bucket: index, MiB, "bwd" (handed to RCCL while the backward pass was still running) or "end", milliseconds before (-) / after the closing call
        self.last_bucket_log = [dict(bucket=i, mib=mib, when="bwd" if early else "end", ms=round((t - t_close) * 1e3, 2)) for i, mib, early, t in self._ov.get("log", [])]
        self._ov = None
        return world


class HipAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) as one fused HIP kernel per
    parameter group over the flat arena; also refreshes the bf16 compute shadow in the same pass."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.arena = ParamArena(self.param_groups)
        self.exp_avg = torch.zeros_like(self.arena.master)
        self.exp_avg_sq = torch.zeros_like(self.arena.master)
        self._step = 0
        self.grad_scale = 1.0  # set by the trainer: 1/world x clip coefficient

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._step += 1
        a = self.arena
        for g, seg in zip(self.param_groups, a.groups):
            s, e = seg["start"], seg["end"]
            if e == s:
                continue
            b1, b2 = g["betas"]
            ops.adamw_step_(a.master[s:e], a.grad[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], a.shadow[s:e], g["lr"], b1, b2,
                            g["eps"], g["weight_decay"], self._step, self.grad_scale)
        from .functional import bump_weight_version, refresh_transposes

        bump_weight_version()
        refresh_transposes(a)   # the transposed weight copies of the next backward, one launch
        return loss

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _layout(self):
        """(offset, numel) of every arena parameter in param_groups order: travels with the flat moments so that a reader with another layout notices"""
        return [(int(p._antmmf_offset), int(p.numel())) for g in self.param_groups for p in g["params"] if getattr(p, "_antmmf_arena", None) is self.arena]

    def state_dict(self):
        d = super().state_dict()
        d["antmmf_arena"] = dict(step=self._step, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, layout=self._layout())
        return d

    def load_state_dict(self, state_dict):
        """Accepts this optimizer's own state (flat moments under "antmmf_arena") AND a reference-format torch.optim.AdamW state_dict
        (per-parameter exp_avg / exp_avg_sq / step, ids in param_groups order -- what a released AntMMF checkpoint holds,
        antmmf/common/checkpoint.py:229-238): those moments are scattered into the flat buffers at the parameters' arena offsets."""
        state_dict = dict(state_dict)  # the caller's dict is left untouched
        extra = state_dict.pop("antmmf_arena", None)
        per_param = state_dict.get("state", {}) or {}
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if extra is not None:
            self._step = int(extra["step"])
            mine = self._layout()
            theirs = extra.get("layout", None)
            if theirs is None:   # written before round 6: the arena order was the parameter order, without pack reordering
                theirs = [(self.arena.legacy_offsets[id(p)], int(p.numel())) for g in self.param_groups for p in g["params"] if getattr(p, "_antmmf_arena", None) is self.arena]
            theirs = [tuple(t) for t in theirs]
            if theirs == mine:
                self.exp_avg.copy_(extra["exp_avg"])
                self.exp_avg_sq.copy_(extra["exp_avg_sq"])
                return
            if len(theirs) != len(mine) or any(a[1] != b[1] for a, b in zip(theirs, mine)) or extra["exp_avg"].numel() != self.exp_avg.numel():
                raise ValueError("HipAdamW.load_state_dict: the flat optimizer state belongs to another set of parameters")
            for (so, n), (do, _) in zip(theirs, mine):       # same parameters, another arena order: moments move parameter by parameter
                self.exp_avg[do:do + n].copy_(extra["exp_avg"][so:so + n])
                self.exp_avg_sq[do:do + n].copy_(extra["exp_avg_sq"][so:so + n])
            return
        if not per_param:
            return
        idx, loaded, step = 0, 0, 0
        for g in self.param_groups:
            for p in g["params"]:
                st = per_param.get(idx, per_param.get(str(idx)))
                idx += 1
                off = getattr(p, "_antmmf_offset", None)
                if st is None or off is None or getattr(p, "_antmmf_arena", None) is not self.arena:
                    continue
                n = p.numel()
                if st["exp_avg"].numel() != n:
                    raise ValueError(f"optimizer state of parameter {idx - 1} has {st['exp_avg'].numel()} elements, expected {n}")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(st["step"]))
                loaded += 1
        if loaded == 0:
            raise ValueError("HipAdamW.load_state_dict: the state matches none of this optimizer's parameters")
        self._step = step
