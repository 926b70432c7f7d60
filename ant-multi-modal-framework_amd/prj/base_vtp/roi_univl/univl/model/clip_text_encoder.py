"""RobertBertEncoder / BertModel2: the in-repo BERT registered into the TextEncoder family (reference:
prj/base_vtp/roi_univl/univl/model/clip_text_encoder.py:25-263).  Contract kept: forward(input_ids,
attention_mask, ...) -> (sequence_output [B,N,d], pooled [B,out_dim]); attributes embeddings, encoder, module,
text_projection, out_dim."""
import os

import torch
from torch import nn

from antmmf.hip import functional as HF
from antmmf.modules.encoders import TextEncoder
from antmmf.modules.vision.backbone.clip.configuration_bert import BertConfig
from antmmf.modules.vision.backbone.clip.modeling_bert import BertModel


class BertModel2(BertModel):
    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        # additive key bias: 0 where attended, -10000 where masked (reference :86-100), kept per key: [B, N] fp32
        key_bias = (1.0 - attention_mask.float()) * -10000.0
        x = self.embeddings(input_ids, token_type_ids=token_type_ids, position_ids=position_ids)
        seq = self.encoder(x, key_bias, head_mask=None)[0]
        return seq, seq[:, 0, :]


@TextEncoder.register()
class RobertBertEncoder(nn.Module):
    def __init__(self, model_name: str = "ViT-B-16", pretrained: bool = True, num_segments: int = None, model_type: str = "bert",
                 bert_model_name: str = "roberta_chinese_base", hidden_size: int = 768, intermediate_size: int = 3072,
                 num_hidden_layers: int = 12, start_hidden_layer: int = 0, num_attention_heads: int = 12,
                 output_attentions: bool = False, output_hidden_states: bool = False, vocab_size: int = 30522,
                 gradient_checkpointing: bool = False, type_vocab_size: int = 2, max_position_embeddings: int = 512,
                 hidden_act: str = "gelu", hidden_dropout_prob: float = 0.1, attention_probs_dropout_prob: float = 0.1,
                 initializer_range: float = 0.02, layer_norm_eps: float = 1e-6, is_proj: bool = True, out_dim: int = 768):
        super().__init__()
        # the reference hard-wires layer_norm_eps = 1e-12 regardless of the argument (clip_text_encoder.py:176)
        self.bert_config = BertConfig(
            vocab_size_or_config_json_file=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
            num_attention_heads=num_attention_heads, intermediate_size=intermediate_size, hidden_act=hidden_act,
            hidden_dropout_prob=hidden_dropout_prob, attention_probs_dropout_prob=attention_probs_dropout_prob,
            max_position_embeddings=max_position_embeddings, type_vocab_size=type_vocab_size,
            initializer_range=initializer_range, layer_norm_eps=1e-12)
        module = BertModel2(self.bert_config)
        self.encoder = module.encoder
        self.embeddings = module.embeddings
        self.module = module
        self.out_dim = out_dim
        self.num_segments = num_segments
        self.text_projection = nn.Parameter(torch.empty(hidden_size, out_dim)) if is_proj else None
        if self.text_projection is not None:
            # the reference leaves torch.empty() uninitialised when pretrained=False (:187-190); CLIP's init is used here
            nn.init.normal_(self.text_projection, std=hidden_size ** -0.5)
        if pretrained:
            self.load_pretrained(model_name)

    def load_pretrained(self, name):
        if not os.path.isfile(name):
            raise RuntimeError(f"Model {name} not found (no network here: pass a local checkpoint path or pretrained=False)")
        sd = torch.load(name, map_location="cpu")["state_dict"]
        picked = {}
        for k, v in sd.items():
            if "text_projection" not in k and "bert" not in k:
                continue
            k = k[len("module."):] if k.startswith("module.") else k
            picked[k[len("bert."):] if k.startswith("bert.") else k] = v
        proj = picked.pop("text_projection", None)
        self.module.load_state_dict(picked, strict=False)
        if proj is not None and self.text_projection is not None and proj.shape == self.text_projection.shape:
            self.text_projection.data.copy_(proj)

    def forward(self, input_ids, attention_mask, token_type_ids=None, position_ids=None, head_mask=None, output_attentions=False):
        seq, cls = self.module(input_ids, attention_mask, token_type_ids, position_ids, head_mask)
        if self.text_projection is not None:
            return seq, HF.linear(cls.contiguous(), self.text_projection, weight_layout="io")
        return seq, cls
