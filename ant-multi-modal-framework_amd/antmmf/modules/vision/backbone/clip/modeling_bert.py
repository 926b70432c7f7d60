"""In-repo BERT on the MI355X kernels: same classes / parameter names as the reference's
antmmf/modules/vision/backbone/clip/modeling_bert.py:66-534 (BertSelfAttention, BertSelfOutput, BertAttention,
BertIntermediate, BertOutput, BertLayer, BertEncoder, BertPooler, BertModel) so state_dicts map 1:1.  Each
BertLayer is one fused autograd node (antmmf.hip.functional.transformer_layer, kind "bert": post-LN, erf-GELU,
additive -10000 key mask).  The sub-modules are parameter holders; their own forward is not on the path.

Dropout: in training mode the fused layer applies attention-probability dropout inside the attention kernels and hidden dropout
after both dense layers, with counter-based masks that forward and backward regenerate (nothing stored); the mask stream is this
build's own (not torch's Philox), so runs are reproducible per seed but not bit-identical to the reference's random draws.
The flagship M2 path has p = 0 everywhere (torchscale config.py:14-17).
"""
import math

import torch
from torch import nn

from antmmf.hip import functional as HF

BertLayerNorm = torch.nn.LayerNorm


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("hidden size must be a multiple of the number of attention heads")
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        # the fused layer consumes q / k / v as ONE [3d, d] operand: ask the optimizer arena to keep them adjacent, in that order (antmmf.hip.arena.tag_pack)
        from antmmf.hip.arena import tag_pack

        tag_pack(self.query.weight, self.key.weight, self.value.weight)
        tag_pack(self.query.bias, self.key.bias, self.value.bias)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size // config.num_attention_heads != 64:
            raise ValueError("the fused attention kernel is specialised for head_dim 64")
        if config.hidden_act != "gelu":
            raise NotImplementedError("BERT on this path uses the erf GELU (modeling_bert.py:31-37)")
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._spec = HF.LayerSpec(kind="bert", heads=config.num_attention_heads, eps=config.layer_norm_eps, act="gelu", packed_qkv=False)
        # training-mode twin with the two dropouts of the block (attention probabilities, both dense outputs) switched on
        self._spec_train = HF.LayerSpec(kind="bert", heads=config.num_attention_heads, eps=config.layer_norm_eps, act="gelu", packed_qkv=False,
                                        attn_dropout=float(config.attention_probs_dropout_prob), hidden_dropout=float(config.hidden_dropout_prob))

    def _params(self):
        a, o = self.attention, self.output
        return dict(wq=a.self.query.weight, bq=a.self.query.bias, wk=a.self.key.weight, bk=a.self.key.bias,
                    wv=a.self.value.weight, bv=a.self.value.bias, wo=a.output.dense.weight, bo=a.output.dense.bias,
                    ln1_w=a.output.LayerNorm.weight, ln1_b=a.output.LayerNorm.bias,
                    w1=self.intermediate.dense.weight, b1=self.intermediate.dense.bias,
                    w2=o.dense.weight, b2=o.dense.bias, ln2_w=o.LayerNorm.weight, ln2_b=o.LayerNorm.bias)

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        """hidden_states [B, N, d] bf16; attention_mask: additive key bias, [B, N] fp32 or the reference's
        extended [B, 1, 1, N] form."""
        if head_mask is not None:
            raise NotImplementedError("head_mask is not supported on the HIP path")
        kb = None
        if attention_mask is not None:
            kb = attention_mask.reshape(attention_mask.shape[0], -1).float().contiguous()
        return HF.transformer_layer(hidden_states, self._spec_train if self.training else self._spec, self._params(), kb)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.grad_checkpointing = False
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, head_mask=None, output_attentions=None):
        """output_attentions (constructor flag or call argument): the attention MAPS of the reference ([B, heads, N, N] per layer, modeling_bert.py:283-314)
        never reach HBM on the fused path.  What comes back in their place is the reduction their only consumer on this path applies
        (UnivlVideoBase.forward_text_encoder, univl_video_base.py:138-143): `KeyImportance` = sum over layers of the head-mean attention summed over the
        queries, a [B, N] fp32 tensor gathered by the attention kernels' sibling antmmf_attention_key_importance (post-dropout probabilities, as HF returns)."""
        want_att = self.output_attentions if output_attentions is None else bool(output_attentions)
        all_hidden = ()
        importance = None
        if want_att:
            importance = torch.zeros(hidden_states.shape[:2], dtype=torch.float32, device=hidden_states.device)
            HF.KEY_IMPORTANCE = importance
        try:
            for layer in self.layer:
                if self.output_hidden_states:
                    all_hidden = all_hidden + (hidden_states,)
                hidden_states = layer(hidden_states, attention_mask, None)
        finally:
            HF.KEY_IMPORTANCE = None
        outputs = (hidden_states,)
        if self.output_hidden_states:
            outputs = outputs + (all_hidden + (hidden_states,),)
        if want_att:
            outputs = outputs + (KeyImportance(importance),)
        return outputs


class KeyImportance:
    """Stands where the tuple of attention maps would: `.sum` of the maps' head means over layers and queries ([B, N])."""

    def __init__(self, value):
        self.value = value


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return torch.tanh(HF.linear(hidden_states[:, 0].contiguous(), self.dense.weight, self.dense.bias).float())


class BertEmbeddings(nn.Module):
    """word + position + token_type lookup -> LayerNorm (-> dropout)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids=None, inputs_embeds=None, token_type_ids=None, position_ids=None):
        """Token ids with all-zero token types take the fused gather kernel.  The stage-2 cross encoder's other uses
        (univl_video_base.py:168-204) -- pre-computed `inputs_embeds` (clip tokens + the [SEP] word embedding) and token type 1 --
        are a few [B, n + 1, d] elementwise adds in front of the same LayerNorm kernel."""
        if position_ids is not None:
            raise NotImplementedError("explicit position_ids (default positions 0..N-1 only)")
        nonzero_types = token_type_ids is not None and bool((token_type_ids != 0).any())
        if inputs_embeds is None and not nonzero_types:
            x = HF.embed(input_ids, self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight, padding_idx=0)
        else:
            if inputs_embeds is None:
                x = HF.embed(input_ids, self.word_embeddings.weight, self.position_embeddings.weight, None, padding_idx=0).float()
                seq = input_ids.shape[1]
            else:
                seq = inputs_embeds.shape[1]
                x = inputs_embeds.float() + self.position_embeddings.weight[:seq][None]
            if token_type_ids is None:
                token_type_ids = torch.zeros(x.shape[:2], dtype=torch.long, device=x.device)
            x = (x + self.token_type_embeddings.weight[token_type_ids]).to(torch.bfloat16)
        x = HF.layer_norm(x, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        if self.training and self.dropout.p > 0:
            x = self.dropout(x)
        return x


class BertModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()
