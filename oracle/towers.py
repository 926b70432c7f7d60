"""oracle.towers -- CPU restatement (plain torch fp32, functional, state-dict driven) of the three
transformer families on the contrastive hot path.  TEST INFRASTRUCTURE ONLY (see oracle/ops.py).

Parameters are looked up in a flat dict `P` under the reference's own state_dict names plus a
prefix, so fixtures / weights map 1:1 onto the reference modules.
"""
import torch

from . import ops


def _sub(P, prefix):
    """View of P with `prefix` stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


# ------------------------------------------------------------------------------ CLIP ViT
def clip_block(P, x, heads):
    """ResidualAttentionBlock on [B, N, d]  (antmmf/modules/vision/backbone/clip/model.py:227-256):
    x += MHA(LN1(x));  x += c_proj(QuickGELU(c_fc(LN2(x)))).  LayerNorm eps 1e-5 (nn.LayerNorm default)."""
    h = ops.layer_norm(x, P["ln_1.weight"], P["ln_1.bias"], 1e-5)
    x = x + ops.clip_mha(h, P["attn.in_proj_weight"], P["attn.in_proj_bias"],
                         P["attn.out_proj.weight"], P["attn.out_proj.bias"], heads)
    h = ops.layer_norm(x, P["ln_2.weight"], P["ln_2.bias"], 1e-5)
    u = ops.linear(h, P["mlp.c_fc.weight"], P["mlp.c_fc.bias"])
    return x + ops.linear(ops.quick_gelu(u), P["mlp.c_proj.weight"], P["mlp.c_proj.bias"])


def dmae_block(P, x, key_bias, heads):
    """ResidualAttentionBlockDmae on [B, N, d] (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:589-611): the CLIP block with
    TF-style LayerNorm eps 1e-12 (:574-587) and an additive attention mask; the mask the caller builds depends on the key
    only ((1 - video_mask) * -1e6 expanded over queries, :205-206), so it is restated as a per-key bias [B, N]."""
    h = ops.layer_norm(x, P["ln_1.weight"], P["ln_1.bias"], 1e-12)
    x = x + ops.clip_mha(h, P["attn.in_proj_weight"], P["attn.in_proj_bias"],
                         P["attn.out_proj.weight"], P["attn.out_proj.bias"], heads, key_bias)
    h = ops.layer_norm(x, P["ln_2.weight"], P["ln_2.bias"], 1e-12)
    u = ops.linear(h, P["mlp.c_fc.weight"], P["mlp.c_fc.bias"])
    return x + ops.linear(ops.quick_gelu(u), P["mlp.c_proj.weight"], P["mlp.c_proj.bias"])


def dmae_agg_visual_feat(P, visual_output, video_mask, heads, layers, sim_header="seqTransf"):
    """DmaeUtils._agg_visual_feat (dmae_utils.py:186-227) for one token per frame (expand_times = 1):
    seqTransf: v + frame_position_embeddings(arange(n)) -> `layers` x dmae_block with key bias (1 - mask) * -1e6 -> + v.
    Returns (visual_output, video_token_mask, visual_output_original)."""
    if sim_header == "meanP":
        return visual_output, video_mask, visual_output
    n = visual_output.shape[1]
    x = visual_output + P["frame_position_embeddings.weight"][:n][None]
    key_bias = (1.0 - video_mask.float()) * -1000000.0
    for i in range(layers):
        x = dmae_block(_sub(P, f"transformerClip.resblocks.{i}."), x, key_bias, heads)
    return x + visual_output, video_mask, visual_output


def patchify(image, patch):
    """[B, C, H, W] -> [B, G*G, C*patch*patch], inner order (c, py, px) == Conv2d weight.flatten(1)."""
    b, c, hh, ww = image.shape
    gh, gw = hh // patch, ww // patch
    x = image.view(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, gh * gw, c * patch * patch)


def clip_vit(P, image, heads, patch):
    """VisionTransformer.forward (clip/model.py:309-335): bias-free patch conv as a GEMM, [cls]+pos,
    ln_pre, L blocks, ln_post(cls) @ proj."""
    w = P["conv1.weight"]
    x = patchify(image, patch) @ w.flatten(1).t()
    cls = P["class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + P["positional_embedding"]
    x = ops.layer_norm(x, P["ln_pre.weight"], P["ln_pre.bias"], 1e-5)
    i = 0
    while f"transformer.resblocks.{i}.ln_1.weight" in P:
        x = clip_block(_sub(P, f"transformer.resblocks.{i}."), x, heads)
        i += 1
    x = ops.layer_norm(x[:, 0], P["ln_post.weight"], P["ln_post.bias"], 1e-5)
    if "proj" in P and P["proj"] is not None:
        x = x @ P["proj"]
    return x


# ------------------------------------------------------------------------------ BERT
def bert_embeddings(P, input_ids=None, inputs_embeds=None, token_type_ids=None, eps=1e-12):
    """word + position + token-type lookup, LayerNorm (dropout omitted: p=0 / eval).
    prj/base_vtp/roi_univl/univl/model/clip_text_encoder.py:36-60."""
    if inputs_embeds is None:
        inputs_embeds = P["word_embeddings.weight"][input_ids]
    b, n = inputs_embeds.shape[:2]
    pos = P["position_embeddings.weight"][:n][None]
    if token_type_ids is None:
        token_type_ids = torch.zeros(b, n, dtype=torch.long)
    typ = P["token_type_embeddings.weight"][token_type_ids]
    return ops.layer_norm(inputs_embeds + pos + typ, P["LayerNorm.weight"], P["LayerNorm.bias"], eps)


def bert_layer(P, x, key_bias, heads, eps=1e-12, drop=None):
    """Post-LN BERT layer (modeling_bert.py:134-270): self-attention with additive key mask,
    LN(dense(ctx) + x), LN(dense(gelu(dense(y))) + y).
    drop (training with dropout): dict of MULTIPLIERS (mask / (1 - p)) applied where the reference applies nn.Dropout --
    "attn" [B, h, N, N] on the attention probabilities (:157), "hid1" / "hid2" [B, N, d] on the two dense outputs before the
    residual add (:183, :235)."""
    q = ops.linear(x, P["attention.self.query.weight"], P["attention.self.query.bias"])
    k = ops.linear(x, P["attention.self.key.weight"], P["attention.self.key.bias"])
    v = ops.linear(x, P["attention.self.value.weight"], P["attention.self.value.bias"])
    dh = x.shape[-1] // heads
    if drop is None:
        ctx = ops.attention_core(ops.split_heads(q, heads), ops.split_heads(k, heads), ops.split_heads(v, heads), dh ** -0.5, key_bias)
    else:
        s = torch.matmul(ops.split_heads(q, heads), ops.split_heads(k, heads).transpose(-1, -2)) * dh ** -0.5
        if key_bias is not None:
            s = s + key_bias[:, None, None, :]
        ctx = torch.matmul(torch.softmax(s, dim=-1) * drop["attn"], ops.split_heads(v, heads))
    ctx = ops.merge_heads(ctx)
    d1 = ops.linear(ctx, P["attention.output.dense.weight"], P["attention.output.dense.bias"])
    a = (d1 if drop is None else d1 * drop["hid1"]) + x
    a = ops.layer_norm(a, P["attention.output.LayerNorm.weight"], P["attention.output.LayerNorm.bias"], eps)
    u = ops.gelu_erf(ops.linear(a, P["intermediate.dense.weight"], P["intermediate.dense.bias"]))
    d2 = ops.linear(u, P["output.dense.weight"], P["output.dense.bias"])
    o = (d2 if drop is None else d2 * drop["hid2"]) + a
    return ops.layer_norm(o, P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], eps)


def bert_key_bias(mask):
    """(1 - mask) * -10000 per key  (clip_text_encoder.py:86-100)."""
    return (1.0 - mask.float()) * -10000.0


def bert_encoder(P, x, key_bias, heads):
    i = 0
    while f"layer.{i}.attention.self.query.weight" in P:
        x = bert_layer(_sub(P, f"layer.{i}."), x, key_bias, heads)
        i += 1
    return x


def roberta_bert_encoder(P, input_ids, attention_mask, heads):
    """RobertBertEncoder.forward (clip_text_encoder.py:248-263): (sequence_output, cls @ text_projection)."""
    x = bert_embeddings(_sub(P, "embeddings."), input_ids)
    x = bert_encoder(_sub(P, "encoder."), x, bert_key_bias(attention_mask), heads)
    pooled = x[:, 0]
    if P.get("text_projection") is not None:
        pooled = pooled @ P["text_projection"]
    return x, pooled


# ------------------------------------------------------------------------------ M2 (BEiT-3 multiway)
def m2_layer(P, x, branch, heads, pad=None, eps=1e-5):
    """torchscale EncoderLayer, pre-LN + sub-LN, one multiway branch ("A" vision / "B" text) for the
    whole tensor (prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:113-168,
    component/multihead_attention.py:66-154, feedforward_network.py:117-128, multiway_network.py:33-45)."""
    br = branch
    h = ops.layer_norm(x, P[f"self_attn_layer_norm.{br}.weight"], P[f"self_attn_layer_norm.{br}.bias"], eps)
    dh = x.shape[-1] // heads
    q = ops.linear(h, P[f"self_attn.q_proj.{br}.weight"], P[f"self_attn.q_proj.{br}.bias"]) * dh ** -0.5
    k = ops.linear(h, P[f"self_attn.k_proj.{br}.weight"], P[f"self_attn.k_proj.{br}.bias"])
    v = ops.linear(h, P[f"self_attn.v_proj.{br}.weight"], P[f"self_attn.v_proj.{br}.bias"])
    key_bias = None
    if pad is not None:
        key_bias = torch.zeros(pad.shape, dtype=torch.float32).masked_fill(pad.bool(), float("-inf"))
    ctx = ops.merge_heads(ops.attention_core(ops.split_heads(q, heads), ops.split_heads(k, heads),
                                             ops.split_heads(v, heads), 1.0, key_bias))
    ctx = ops.layer_norm(ctx, P[f"self_attn.inner_attn_ln.{br}.weight"], P[f"self_attn.inner_attn_ln.{br}.bias"], eps)
    x = x + ops.linear(ctx, P[f"self_attn.out_proj.{br}.weight"], P[f"self_attn.out_proj.{br}.bias"])
    h = ops.layer_norm(x, P[f"final_layer_norm.{br}.weight"], P[f"final_layer_norm.{br}.bias"], eps)
    u = ops.gelu_erf(ops.linear(h, P[f"ffn.{br}.fc1.weight"], P[f"ffn.{br}.fc1.bias"]))
    u = ops.layer_norm(u, P[f"ffn.{br}.ffn_layernorm.weight"], P[f"ffn.{br}.ffn_layernorm.bias"], eps)
    return x + ops.linear(u, P[f"ffn.{br}.fc2.weight"], P[f"ffn.{br}.fc2.bias"])


def m2_encoder(P, x, branch, heads, pad=None, pos_branch=None):
    """torchscale Encoder.forward (encoder.py:388-482) restricted to what ITC uses: optional multiway
    positional embedding (positions start at 2, embedding.py:92-110), zeroing of padded positions
    (encoder.py:440), L layers, final LayerNorm."""
    if pos_branch is not None:
        n = x.shape[1]
        x = x + P[f"embed_positions.{pos_branch}.weight"][2:n + 2][None]
    if pad is not None:
        x = x * (1 - pad[..., None].to(x.dtype))
    i = 0
    while f"layers.{i}.self_attn.q_proj.A.weight" in P:
        x = m2_layer(_sub(P, f"layers.{i}."), x, branch, heads, pad)
        i += 1
    return ops.layer_norm(x, P[f"layer_norm.{branch}.weight"], P[f"layer_norm.{branch}.bias"], 1e-5)


def m2_infer_image(P, image, heads, patch):
    """VLMo.infer_image (prj/M2_Encoder/vlmo/modules/vlmo_module.py:359-405): (x-.5)/.5, conv patch
    embed with bias + cls, backbone (branch A), backbone_vl (branch A), two bias-free heads, x/||x||."""
    x = (image - 0.5) / 0.5
    w, b = P["backbone.vision_embed.proj.weight"], P["backbone.vision_embed.proj.bias"]
    x = patchify(x, patch) @ w.flatten(1).t() + b
    cls = P["backbone.vision_embed.cls_token"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    vffn = m2_encoder(_sub(P, "backbone.encoder."), x, "A", heads, None, pos_branch="A")
    vlffn = m2_encoder(_sub(P, "backbone_vl."), vffn, "A", heads, None)
    f1 = ops.l2_normalize_noeps(vffn[:, 0] @ P["itc_image_proj.fc.weight"].t())
    f2 = ops.l2_normalize_noeps(vlffn[:, 0] @ P["itc_vl_image_proj.fc.weight"].t())
    return dict(image_feats=vffn, cls_feats=f1, cls_vlffn_feats=f2)


def m2_infer_text(P, text_ids, text_masks, heads):
    """VLMo.infer_text (vlmo_module.py:323-357): embedding lookup, backbone with the text branch (B)
    and key-padding mask, backbone_vl with branch **A** (multiway_split_position=-1, :338-343)."""
    pad = (1 - text_masks).bool()
    x = P["backbone.text_embed.weight"][text_ids]
    lffn = m2_encoder(_sub(P, "backbone.encoder."), x, "B", heads, pad, pos_branch="B")
    vlffn = m2_encoder(_sub(P, "backbone_vl."), lffn, "A", heads, pad)
    f1 = ops.l2_normalize_noeps(lffn[:, 0] @ P["itc_text_proj.fc.weight"].t())
    f2 = ops.l2_normalize_noeps(vlffn[:, 0] @ P["itc_vl_text_proj.fc.weight"].t())
    return dict(cls_feats=f1, cls_vlffn_feats=f2)
