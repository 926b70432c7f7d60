"""Size presets of the M2 towers (reference: prj/M2_Encoder/vlmo/modules/modeling_utils.py:21-141)."""
from ..torchscale.architecture.config import EncoderConfig
from ..torchscale.model.BEiT3 import BEiT3

_PRESETS = {"base": dict(encoder_layers=12, encoder_embed_dim=768, encoder_attention_heads=12, vocab_size=64010),
            "large": dict(encoder_layers=24, encoder_embed_dim=1024, encoder_attention_heads=16, vocab_size=64010),
            "huge": dict(encoder_layers=32, encoder_embed_dim=4096, encoder_attention_heads=32, vocab_size=30522)}


def get_config(version, img_size=224, patch_size=16, drop_path_rate=0, checkpoint_activations=None, mlp_ratio=4, **kw):
    p = dict(_PRESETS[version])
    for k in ("vocab_size", "encoder_layers", "encoder_embed_dim", "encoder_attention_heads"):
        if kw.get(k) is not None:
            p[k] = kw[k]
    return EncoderConfig(img_size=img_size, patch_size=patch_size, vocab_size=p["vocab_size"], multiway=True, layernorm_embedding=False,
                         normalize_output=True, no_output_layer=True, drop_path_rate=drop_path_rate,
                         encoder_embed_dim=p["encoder_embed_dim"], encoder_attention_heads=p["encoder_attention_heads"],
                         encoder_layers=p["encoder_layers"], encoder_ffn_embed_dim=int(p["encoder_embed_dim"] * mlp_ratio),
                         checkpoint_activations=bool(checkpoint_activations), share_layer=kw.get("share_layer", False),
                         share_attn=kw.get("share_attn", False), deepnorm=kw.get("deepnorm", False), mask_ratio=kw.get("mask_ratio", 0),
                         max_text_len=kw.get("max_text_len", 52), one_attn=kw.get("one_attn", False))


def _get_base_config(**kw):
    return get_config("base", **kw)


def _get_large_config(**kw):
    return get_config("large", **kw)


def _get_huge_config(**kw):
    return get_config("huge", **kw)


__all__ = ["BEiT3", "get_config", "_get_base_config", "_get_large_config", "_get_huge_config"]
