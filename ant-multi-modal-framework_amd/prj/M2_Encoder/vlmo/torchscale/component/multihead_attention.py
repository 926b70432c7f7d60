"""MultiheadAttention parameter holder: separate q/k/v/out projections per multiway branch + inner_attn_ln
(reference: prj/M2_Encoder/vlmo/torchscale/component/multihead_attention.py:19-154)."""
import math

from torch import nn

from .multiway_network import MultiwayWrapper


class MultiheadAttention(nn.Module):
    def __init__(self, args, embed_dim, num_heads, dropout=0.0, self_attention=False, encoder_decoder_attention=False, subln=False, one_attn=False):
        super().__init__()
        assert self_attention and not encoder_decoder_attention
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        if self.head_dim != 64:
            raise ValueError("the fused attention kernel is specialised for head_dim 64 (all M2 sizes)")
        self.scaling = self.head_dim ** -0.5
        self.k_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.v_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.q_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.out_proj = MultiwayWrapper(args, nn.Linear(embed_dim, embed_dim, bias=True))
        self.inner_attn_ln = MultiwayWrapper(args, nn.LayerNorm(embed_dim, eps=args.layernorm_eps)) if subln else None
        # the fused layer consumes q / k / v as ONE [3d, d] operand: ask the optimizer arena to keep them adjacent, in that order (antmmf.hip.arena.tag_pack)
        from antmmf.hip.arena import tag_pack

        for br in ("A", "B"):
            q, k, v = (getattr(self, nm).pick(br) for nm in ("q_proj", "k_proj", "v_proj"))
            if getattr(q.weight, "_antmmf_pack", None) is None:     # (without multiway both branches are the same module)
                tag_pack(q.weight, k.weight, v.weight)
                tag_pack(q.bias, k.bias, v.bias)

    def reset_parameters(self):
        for br in ("A", "B"):
            for nm, gain in (("q_proj", 1 / math.sqrt(2)), ("k_proj", 1 / math.sqrt(2)), ("v_proj", 1 / math.sqrt(2)), ("out_proj", 1.0)):
                lin = getattr(self, nm).pick(br)
                nn.init.xavier_uniform_(lin.weight, gain=gain)
            nn.init.constant_(self.out_proj.pick(br).bias, 0.0)
