"""`PretrainedTransformerEncoder`: the text / cross tower every shipped *_vtp yml with `arch_type: univl` names
(reference antmmf/modules/encoders/text_encoder.py:32-175: HuggingFace `AutoModel.from_pretrained(bert dir)`, truncated to
`encoder.layer[start : start + num_hidden_layers]`, token-type table resized to `num_segments`).

SURVEY.md 7: HuggingFace is a weight-LAYOUT source here, not a runtime: the parameters live in the in-repo fused BERT
(antmmf/modules/vision/backbone/clip/modeling_bert.py -> hip.functional.transformer_layer(kind="bert")), whose module tree carries the
HF BertModel names (embeddings.word_embeddings ... encoder.layer.N.attention.self.query ... pooler.dense), so a released
`pytorch_model.bin` / `model.safetensors` of a BERT checkpoint loads key for key (a leading `bert.` is dropped; heads the model does not
have -- `cls.*` -- are ignored).  Same constructor arguments, same attributes (`embeddings`, `encoder`, `pooler`, `module`, `config`),
same forward triple.  Differences, stated: nothing is downloaded (a missing checkpoint directory is an error); `output_attentions=True`
returns, in the place of the attention maps (which never reach HBM on the fused path), their reduction `KeyImportance` -- the sum over layers of
the head-mean attention summed over the queries, which is all the caller derives from them (`words_importance`, univl_video_base.py:138-143);
`model_type` other than "bert" is out of scope."""
import json
import os
import warnings

import torch
from torch import nn

from antmmf.modules.encoders import TextEncoder
from antmmf.modules.vision.backbone.clip.configuration_bert import BertConfig
from antmmf.modules.vision.backbone.clip.modeling_bert import BertModel

BERT_PRETRAINED_MODELS_ENV_VAR = "PYTORCH_TRANSFORMERS_CACHE"


def _pretrained_dir(name):
    return name if os.path.isabs(name) else os.path.join(os.environ.get(BERT_PRETRAINED_MODELS_ENV_VAR, ""), name)


def load_hf_bert_state_dict(path):
    """A HuggingFace BERT checkpoint directory (or file) -> {in-repo parameter name: tensor}."""
    if os.path.isdir(path):
        for fn in ("model.safetensors", "pytorch_model.bin"):
            if os.path.isfile(os.path.join(path, fn)):
                path = os.path.join(path, fn)
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu")
    out = {}
    for k, v in sd.items():
        k = k[5:] if k.startswith("bert.") else k
        k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")   # TF-era checkpoints
        if k.startswith(("embeddings.", "encoder.", "pooler.")) and not k.endswith("position_ids"):
            out[k] = v
    return out


class _FusedBert(BertModel):
    """BertModel (HF parameter names) with the forward of the HF model the reference calls: -> (sequence_output, pooled_output)."""

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None, inputs_embeds=None,
                output_attentions=False, output_hidden_states=False, return_dict=False):
        if return_dict:
            raise NotImplementedError("PretrainedTransformerEncoder: return_dict=True (the reference calls it with return_dict=False)")
        ref = input_ids if input_ids is not None else inputs_embeds
        if attention_mask is None:
            attention_mask = torch.ones(ref.shape[:2], dtype=torch.long, device=ref.device)
        key_bias = (1.0 - attention_mask.float()) * -10000.0
        x = self.embeddings(input_ids=input_ids, inputs_embeds=inputs_embeds, token_type_ids=token_type_ids, position_ids=position_ids)
        enc = self.encoder(x, key_bias, head_mask=None, output_attentions=output_attentions)
        seq = enc[0]
        if output_attentions:   # third output: modeling_bert.KeyImportance in the place of the attention maps (see BertEncoder.forward)
            return seq, self.pooler(seq), enc[-1]
        return seq, self.pooler(seq)


class PretrainedTransformerEncoderAndEmbedding(nn.Module):
    def __init__(self, pretrained=None, num_segments=None, model_type="bert", bert_model_name="bert-base-uncased", hidden_size=768,
                 intermediate_size=3072, num_hidden_layers=12, start_hidden_layer=0, num_attention_heads=12, output_attentions=False,
                 output_hidden_states=False, vocab_size=30522, gradient_checkpointing=False, type_vocab_size=2, max_position_embeddings=512):
        super().__init__()
        if model_type != "bert":
            raise NotImplementedError(f"PretrainedTransformerEncoder: model_type {model_type!r} (BERT is the text tower of the *_vtp ymls)")
        cfg_kwargs = dict(vocab_size_or_config_json_file=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                          num_attention_heads=num_attention_heads, intermediate_size=intermediate_size, max_position_embeddings=max_position_embeddings,
                          type_vocab_size=type_vocab_size, output_hidden_states=output_hidden_states)
        state = None
        if pretrained is False:
            warnings.warn("random initialization for {}".format(bert_model_name))
            total_layers = start_hidden_layer + num_hidden_layers
        else:
            path = _pretrained_dir(bert_model_name)
            if not os.path.exists(path):
                raise FileNotFoundError(f"PretrainedTransformerEncoder: no checkpoint directory {path!r} for {bert_model_name!r} (set "
                                        f"${BERT_PRETRAINED_MODELS_ENV_VAR}; nothing is downloaded)")
            cj = os.path.join(path, "config.json")
            total_layers = start_hidden_layer + num_hidden_layers
            if os.path.isfile(cj):
                with open(cj) as f:
                    hf = json.load(f)
                # the checkpoint decides the shapes of what is loaded (AutoModel.from_pretrained(path, config=...) merges the same way)
                for ours, theirs in (("vocab_size_or_config_json_file", "vocab_size"), ("hidden_size", "hidden_size"), ("num_attention_heads", "num_attention_heads"),
                                     ("intermediate_size", "intermediate_size"), ("max_position_embeddings", "max_position_embeddings"),
                                     ("type_vocab_size", "type_vocab_size")):
                    if theirs in hf:
                        cfg_kwargs[ours] = hf[theirs]
                for extra in ("layer_norm_eps", "hidden_act", "hidden_dropout_prob", "attention_probs_dropout_prob"):
                    if extra in hf:
                        cfg_kwargs[extra] = hf[extra]
                total_layers = max(total_layers, int(hf.get("num_hidden_layers", total_layers)))
            state = load_hf_bert_state_dict(path)
        cfg_kwargs["num_hidden_layers"] = total_layers
        self.module = _FusedBert(BertConfig(**cfg_kwargs))
        if state is not None:
            missing, unexpected = self.module.load_state_dict(state, strict=False)
            if missing:
                raise RuntimeError(f"PretrainedTransformerEncoder: checkpoint lacks {len(missing)} parameters, e.g. {missing[:4]}")
        # keep the layers asked for (the reference tails the loaded model the same way)
        self.module.encoder.layer = nn.ModuleList(list(self.module.encoder.layer[start_hidden_layer:start_hidden_layer + num_hidden_layers]))
        self.module.config.num_hidden_layers = len(self.module.encoder.layer)
        self.embeddings = self.module.embeddings
        self.num_segments = num_segments
        self.config = self.module.config
        self._init_segment_embeddings()

    def _init_segment_embeddings(self):
        if self.num_segments is None:
            return
        old = self.embeddings.token_type_embeddings
        if self.num_segments == old.num_embeddings:
            return
        new = nn.Embedding(self.num_segments, self.config.hidden_size)
        new.weight.data[:2].copy_(old.weight.data[:2])
        for idx in range(2, self.num_segments - 1):   # (the reference's range: the last row keeps its fresh initialisation)
            new.weight.data[idx].copy_(old.weight.data.mean(dim=0))
        self.embeddings.token_type_embeddings = new.to(old.weight.device)

    def forward(self, *args, return_sequence=False, **kwargs):
        out = self.module(*args, **kwargs)
        return out[0] if return_sequence else out[1]


@TextEncoder.register()
class PretrainedTransformerEncoder(nn.Module):
    def __init__(self, pretrained=None, num_segments=None, model_type="bert", bert_model_name="bert-base-uncased", hidden_size=768,
                 intermediate_size=3072, num_hidden_layers=12, start_hidden_layer=0, num_attention_heads=12, output_attentions=False,
                 output_hidden_states=False, vocab_size=30522, gradient_checkpointing=False, type_vocab_size=2, max_position_embeddings=512):
        super().__init__()
        module = PretrainedTransformerEncoderAndEmbedding(
            pretrained=pretrained, num_segments=num_segments, model_type=model_type, bert_model_name=bert_model_name, hidden_size=hidden_size,
            intermediate_size=intermediate_size, num_hidden_layers=num_hidden_layers, start_hidden_layer=start_hidden_layer,
            num_attention_heads=num_attention_heads, output_attentions=output_attentions, output_hidden_states=output_hidden_states,
            vocab_size=vocab_size, gradient_checkpointing=gradient_checkpointing, type_vocab_size=type_vocab_size,
            max_position_embeddings=max_position_embeddings).module
        self.encoder = module.encoder
        self.embeddings = module.embeddings
        self.pooler = module.pooler
        self.module = module
        self.config = module.config
        self.text_projection = None   # (clip-arch towers carry one; the univl arch pools with `pooler`)

    def forward(self, *args, return_dict=False, **kwargs):
        return self.module(*args, **kwargs, return_dict=return_dict)
