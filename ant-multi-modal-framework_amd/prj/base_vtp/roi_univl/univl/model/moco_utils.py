"""MocoUtils on the MI355X path: momentum key encoders, negative-key queues, MoCo loss (reference:
prj/base_vtp/roi_univl/univl/model/moco_utils.py:13-107; call site univl_video_ret.py:262-312).

Same module surface (img_encoder_q/_k, txt_encoder_q/_k, img_queue, txt_queue, *_queue_ptr, momentum_update_key_encoder,
moco_loss, dequeue_and_enqueue) and hyper-parameters (K, M, T from the config).  MI355X design:
  * the key towers' parameters are laid out in ONE flat fp32 buffer at the query towers' offsets in the optimizer's
    parameter arena (plus a bf16 compute shadow), so the momentum update is a single fused launch over the whole arena
    (`antmmf_ema_update`: k = m k + (1 - m) q, shadow = bf16(k)) instead of a Python loop over ~190 M parameters;
  * the loss reads the [R, K] negatives once in a fused row kernel (`antmmf_moco_fwd/bwd`);
  * enqueue gathers keys with one all_gather_into_tensor; write pointer (the checkpointed `*_queue_ptr` buffer) and NaN guard
    live on the device: a non-finite batch leaves queue AND pointer untouched, as in the reference, without a host sync."""
import copy

import torch
from torch import nn

from antmmf.hip import contrastive, ops


class MocoUtils(nn.Module):
    def __init__(self, config, img_encoder=None, txt_encoder=None):
        assert img_encoder is not None or txt_encoder is not None
        super().__init__()
        self.config = config
        self.dim = self.config.hidden_size
        self.txt_K = config.get("K", 16384)  # queue size; number of negative keys
        self.img_K = 16384                   # (hard-coded in the reference, moco_utils.py:22)
        self.m = config.get("M", 0.9999)
        self.T = config.get("T", 0.05)
        # the query encoders are referenced, not registered (they belong to the model that owns this object)
        object.__setattr__(self, "img_encoder_q", img_encoder)
        object.__setattr__(self, "txt_encoder_q", txt_encoder)
        self._pairs = []
        if img_encoder is not None:
            self.img_encoder_k = self._key_copy(img_encoder)
            q = torch.nn.functional.normalize(torch.randn(self.dim, self.img_K), dim=0)
            self.register_buffer("img_queue", q)
            self.register_buffer("img_queue_ptr", torch.zeros(1, dtype=torch.long))
        if txt_encoder is not None:
            self.txt_encoder_k = self._key_copy(txt_encoder)
            q = torch.nn.functional.normalize(torch.randn(self.dim, self.txt_K), dim=0)
            self.register_buffer("txt_queue", q)
            self.register_buffer("txt_queue_ptr", torch.zeros(1, dtype=torch.long))
        self._flat = None  # (k_master, k_shadow, q_arena) once the key towers mirror the optimizer arena

    def _key_copy(self, enc_q):
        enc_k = copy.deepcopy(enc_q)  # Parameter.__deepcopy__ clones the data: plain standalone fp32 tensors
        for pq, pk in zip(enc_q.parameters(), enc_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
            self._pairs.append((pq, pk))
        return enc_k

    # ------------------------------------------------------------------ momentum update
    def _mirror_arena(self):
        """Re-home the key parameters at the query parameters' arena offsets (possible once the optimizer built its arena)."""
        arena = getattr(self._pairs[0][0], "_antmmf_arena", None)
        if arena is None or any(getattr(pq, "_antmmf_arena", None) is not arena for pq, _ in self._pairs):
            return False
        k_master = torch.zeros_like(arena.master)
        k_shadow = torch.zeros_like(arena.shadow)
        for pq, pk in self._pairs:
            off, n = pq._antmmf_offset, pq.numel()
            view = k_master[off:off + n].view(pk.shape)
            view.copy_(pk.data)
            pk.data = view
            pk._antmmf_bf16 = k_shadow[off:off + n].view(pk.shape)
        ops.cast_bf16(k_master, out=k_shadow)
        self._flat = (k_master, k_shadow, arena)
        return True

    @torch.no_grad()
    def momentum_update_key_encoder(self):
        if self._flat is None:
            self._mirror_arena()  # cheap no-op until the optimizer has built its parameter arena
        if self._flat is not None:
            k_master, k_shadow, arena = self._flat
            # entries of the arena that are not tower parameters are updated too (unused slots of the key buffer)
            ops.ema_update_(k_master, arena.master, self.m, k_shadow)
            return
        for pq, pk in self._pairs:  # no optimizer arena (unit tests): one launch per parameter
            kd = pk.data.view(-1)
            ops.ema_update_(kd, pq.data.detach().float().contiguous().view(-1), self.m)

    # ------------------------------------------------------------------ loss
    def moco_loss(self, q, kpos, queue):
        """q [R, D]; kpos [R, Np, D]; queue [D, K].  (The reference takes precomputed pos / neg; here they are formed
        inside the fused autograd function so that the [R, K] slab never round-trips through autograd.)"""
        return contrastive.moco_loss(q, kpos, queue, self.T)

    # ------------------------------------------------------------------ queue
    @torch.no_grad()
    def dequeue_and_enqueue(self, vis_keys, txt_keys):
        def push(keys, queue, queue_ptr, K):
            keys = contrastive._all_gather(keys.detach().float().contiguous(), None)
            n = keys.shape[0]
            # the write pointer IS the checkpointed `*_queue_ptr` buffer and stays on the device: the slot indices are computed from it
            # there (end = min(ptr + n, K), start = end - n, as moco_utils.py:98-103), so neither the pointer nor the NaN guard ever
            # syncs with the host.  A non-finite batch leaves queue AND pointer untouched, as in the reference (:92-96).
            start = torch.clamp(queue_ptr + n, max=K) - n
            cols = start + torch.arange(n, device=queue.device)
            ok = torch.isfinite(keys).all()
            new = torch.where(ok, keys.t().to(queue.dtype), queue.index_select(1, cols))
            queue.index_copy_(1, cols, new)
            queue_ptr.copy_(torch.where(ok, (start + n) % K, queue_ptr))

        if self.img_encoder_q is not None:
            push(vis_keys, self.img_queue, self.img_queue_ptr, self.img_K)
        if self.txt_encoder_q is not None:
            push(txt_keys, self.txt_queue, self.txt_queue_ptr, self.txt_K)
