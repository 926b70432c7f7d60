"""torchscale Encoder / EncoderLayer on the MI355X kernels (reference:
prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:28-482), restricted to what the M2 ITC towers use:
pre-LN + sub-LN layers, multiway routing of the whole sequence to branch A (vision) or B (text), key-padding mask,
final LayerNorm.  Each EncoderLayer is one fused autograd node (antmmf.hip.functional.transformer_layer, kind "m2")."""
import math

import torch
from torch import nn

from antmmf.hip import functional as HF
from ..component.feedforward_network import FeedForwardNetwork
from ..component.multihead_attention import MultiheadAttention
from ..component.multiway_network import MultiwayWrapper, branch_of


class EncoderLayer(nn.Module):
    def __init__(self, args, depth, attn=None, is_moe_layer=False, is_encoder_decoder=False):
        super().__init__()
        assert not is_moe_layer and args.multiway
        self.args = args
        self.embed_dim = args.encoder_embed_dim
        self.self_attn = MultiheadAttention(args, self.embed_dim, args.encoder_attention_heads, dropout=args.attention_dropout,
                                            self_attention=True, subln=args.subln, one_attn=args.one_attn)
        self.self_attn_layer_norm = MultiwayWrapper(args, nn.LayerNorm(self.embed_dim, eps=args.layernorm_eps))
        self.ffn_dim = args.encoder_ffn_embed_dim
        self.ffn = MultiwayWrapper(args, FeedForwardNetwork(self.embed_dim, self.ffn_dim, args.activation_fn, args.dropout,
                                                            args.activation_dropout, args.layernorm_eps, args.subln))
        self.final_layer_norm = MultiwayWrapper(args, nn.LayerNorm(self.embed_dim, eps=args.layernorm_eps))
        self.alpha = 1.0
        self._spec = HF.LayerSpec(kind="m2", heads=args.encoder_attention_heads, eps=args.layernorm_eps, act="gelu", packed_qkv=False)

    def _params(self, br):
        a, f = self.self_attn, self.ffn.pick(br)
        ln1, ln2, inner = self.self_attn_layer_norm.pick(br), self.final_layer_norm.pick(br), a.inner_attn_ln.pick(br)
        q, k, v, o = a.q_proj.pick(br), a.k_proj.pick(br), a.v_proj.pick(br), a.out_proj.pick(br)
        return dict(ln1_w=ln1.weight, ln1_b=ln1.bias, wq=q.weight, bq=q.bias, wk=k.weight, bk=k.bias, wv=v.weight, bv=v.bias,
                    inner_w=inner.weight, inner_b=inner.bias, wo=o.weight, bo=o.bias, ln2_w=ln2.weight, ln2_b=ln2.bias,
                    w1=f.fc1.weight, b1=f.fc1.bias, ffn_w=f.ffn_layernorm.weight, ffn_b=f.ffn_layernorm.bias,
                    w2=f.fc2.weight, b2=f.fc2.bias)

    def forward(self, x, encoder_padding_mask=None, attn_mask=None, rel_pos=None, multiway_split_position=None,
                incremental_state=None, key_bias=None):
        if attn_mask is not None or rel_pos is not None or incremental_state is not None:
            raise NotImplementedError("attn_mask / rel_pos / incremental decoding are outside the ITC path")
        br = branch_of(-1 if multiway_split_position is None else multiway_split_position)
        if key_bias is None and encoder_padding_mask is not None:
            key_bias = torch.zeros(encoder_padding_mask.shape, dtype=torch.float32, device=x.device).masked_fill_(
                encoder_padding_mask.bool(), float("-inf"))
        return HF.transformer_layer(x, self._spec, self._params(br), key_bias), None


class Encoder(nn.Module):
    def __init__(self, args, embed_tokens=None, embed_positions=None, output_projection=None, is_encoder_decoder=False, **kwargs):
        super().__init__()
        self.args = args
        embed_dim = args.encoder_embed_dim
        self.embed_scale = 1.0
        self.max_text_len = args.max_text_len
        self.vision_len = (args.img_size // args.patch_size) ** 2
        self.embed_tokens = embed_tokens
        self.embed_positions = embed_positions
        self.output_projection = None
        self.layernorm_embedding = None
        self.layers = nn.ModuleList([EncoderLayer(args, depth=i) for i in range(args.encoder_layers)])
        self.num_layers = len(self.layers)
        self.layer_norm = MultiwayWrapper(args, nn.LayerNorm(embed_dim, eps=args.layernorm_eps))
        self.relative_position = None
        if args.subln:  # sub-LN init: scale fc1 / fc2 / out_proj / v_proj by sqrt(log(2 L)) (reference :257-264)
            init_scale = math.sqrt(math.log(args.encoder_layers * 2))
            for name, p in self.named_parameters():
                if "fc1" in name or "fc2" in name or "out_proj" in name or "v_proj" in name:
                    p.data.mul_(init_scale)

    def forward(self, src_tokens=None, encoder_padding_mask=None, attn_mask=None, return_all_hiddens=False, token_embeddings=None,
                multiway_split_position=None, features_only=False, incremental_state=None, positions=None, pos_added=False, **kwargs):
        """token_embeddings [B, N, d] bf16.  `pos_added` tells that the caller's fused embedding kernel already added
        this encoder's positional embedding and zeroed the padded rows (BEiT3 does)."""
        assert token_embeddings is not None and attn_mask is None and incremental_state is None
        x = token_embeddings
        br = branch_of(-1 if multiway_split_position is None else multiway_split_position)
        if not pos_added:
            if self.embed_positions is not None:
                n = x.shape[1]
                x = x + self.embed_positions.pick(br).weight[2:n + 2].to(x.dtype)[None]
            if encoder_padding_mask is not None:
                x = x * (~encoder_padding_mask.bool())[..., None].to(x.dtype)
        key_bias = None
        if encoder_padding_mask is not None:
            key_bias = torch.zeros(encoder_padding_mask.shape, dtype=torch.float32, device=x.device).masked_fill_(
                encoder_padding_mask.bool(), float("-inf"))
        states = [x] if return_all_hiddens else []
        for layer in self.layers:
            x, _ = layer(x, multiway_split_position=multiway_split_position, key_bias=key_bias)
            if return_all_hiddens:
                states.append(x)
        ln = self.layer_norm.pick(br)
        x = HF.layer_norm(x, ln.weight, ln.bias, ln.eps)
        return {"encoder_out": x, "encoder_embedding": token_embeddings, "encoder_padding_mask": encoder_padding_mask,
                "encoder_states": states, "l_aux": [None] * len(self.layers), "multiway_split_position": multiway_split_position}
