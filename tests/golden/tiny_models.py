"""Shapes of the tiny models behind e2e_clip_arch.pt / e2e_m2.pt, and their name-keyed weights.
Shared by the oracle tests (CPU) and the HIP parity tests (GPU).  Shapes mirror TINY_CLIP_CFG /
TINY_M2 in make_golden.py."""
import torch

import weightgen as W

CLIP_ARCH = dict(width=128, layers=2, heads=2, patch=8, res=32, out_dim=128,
                 vocab=300, hidden=128, inter=512, bert_layers=2, bert_heads=2, max_pos=40)


def clip_arch_shapes(c=CLIP_ARCH):
    d, s = c["width"], {}
    v = "module.img_encoder.visual."
    s[v + "class_embedding"] = (d,)
    s[v + "positional_embedding"] = ((c["res"] // c["patch"]) ** 2 + 1, d)
    s[v + "proj"] = (d, c["out_dim"])
    s[v + "conv1.weight"] = (d, 3, c["patch"], c["patch"])
    for ln in ("ln_pre", "ln_post"):
        s[v + ln + ".weight"] = (d,)
        s[v + ln + ".bias"] = (d,)
    for i in range(c["layers"]):
        b = v + f"transformer.resblocks.{i}."
        s[b + "attn.in_proj_weight"] = (3 * d, d)
        s[b + "attn.in_proj_bias"] = (3 * d,)
        s[b + "attn.out_proj.weight"] = (d, d)
        s[b + "attn.out_proj.bias"] = (d,)
        for ln in ("ln_1", "ln_2"):
            s[b + ln + ".weight"] = (d,)
            s[b + ln + ".bias"] = (d,)
        s[b + "mlp.c_fc.weight"] = (4 * d, d)
        s[b + "mlp.c_fc.bias"] = (4 * d,)
        s[b + "mlp.c_proj.weight"] = (d, 4 * d)
        s[b + "mlp.c_proj.bias"] = (d,)
    t, h = "module.text_encoder.", c["hidden"]
    s[t + "text_projection"] = (h, c["out_dim"])
    s[t + "embeddings.word_embeddings.weight"] = (c["vocab"], h)
    s[t + "embeddings.position_embeddings.weight"] = (c["max_pos"], h)
    s[t + "embeddings.token_type_embeddings.weight"] = (2, h)
    s[t + "embeddings.LayerNorm.weight"] = (h,)
    s[t + "embeddings.LayerNorm.bias"] = (h,)
    for i in range(c["bert_layers"]):
        b = t + f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            s[b + f"attention.self.{nm}.weight"] = (h, h)
            s[b + f"attention.self.{nm}.bias"] = (h,)
        s[b + "attention.output.dense.weight"] = (h, h)
        s[b + "attention.output.dense.bias"] = (h,)
        s[b + "attention.output.LayerNorm.weight"] = (h,)
        s[b + "attention.output.LayerNorm.bias"] = (h,)
        s[b + "intermediate.dense.weight"] = (c["inter"], h)
        s[b + "intermediate.dense.bias"] = (c["inter"],)
        s[b + "output.dense.weight"] = (h, c["inter"])
        s[b + "output.dense.bias"] = (h,)
        s[b + "output.LayerNorm.weight"] = (h,)
        s[b + "output.LayerNorm.bias"] = (h,)
    # BertModel2 also owns a pooler (unused by the clip arch) -- it gets no gradient and is not needed here
    return s


def clip_arch_params(requires_grad=False, stage2=False, dmae=False):
    shapes = clip_arch_shapes()
    if dmae:  # DmaeUtils' weight heads (dmae_utils.py:36-38), meanP header
        h = CLIP_ARCH["hidden"]
        shapes.update({"dmae_utils.text_weight_fc.weight": (1, h), "dmae_utils.text_weight_fc.bias": (1,),
                       "dmae_utils.video_weight_fc.weight": (1, h), "dmae_utils.video_weight_fc.bias": (1,)})
    if stage2:  # similarity_dense of the stage-2 head (univl_video_ret.py:23-29)
        h = CLIP_ARCH["hidden"]
        shapes.update({"similarity_dense.0.weight": (2 * h, h), "similarity_dense.0.bias": (2 * h,),
                       "similarity_dense.2.weight": (1, 2 * h), "similarity_dense.2.bias": (1,)})
    P = W.fill_dict(shapes)
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P


M2 = dict(d=128, heads=2, layers=2, vl_layers=1, patch=8, res=32, vocab=300, out=64, max_src_pos=1024)


def m2_shapes(c=M2):
    from oracle.shapes import m2_shapes as _shapes

    return _shapes(d=c["d"], layers=c["layers"], vl_layers=c["vl_layers"], patch=c["patch"], res=c["res"], vocab=c["vocab"],
                   out=c["out"], max_src_pos=c["max_src_pos"])


def m2_params(expected_names=None, requires_grad=False):
    shapes = m2_shapes()
    if expected_names is not None:
        missing = [n for n in expected_names if n not in shapes and not n.startswith(("norm.", "pooler."))]
        assert not missing, f"tiny M2 shape table is missing reference parameters: {missing[:8]}"
    P = W.fill_dict(shapes)
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P
