"""Parameter-shape tables of the towers (reference state_dict names).  TEST INFRASTRUCTURE (oracle side): used to
instantiate the functional oracle at any size -- tiny parity models, and the M2-large sample that bench.py's
cpu_baseline leg times."""


def m2_shapes(d=128, layers=2, vl_layers=1, patch=8, res=32, vocab=300, out=64, max_src_pos=1024):
    s = {}
    s["logit_scale"] = ()
    s["logit_vl_scale"] = ()
    s["backbone.text_embed.weight"] = (vocab, d)
    s["backbone.vision_embed.mask_token"] = (1, 1, d)
    s["backbone.vision_embed.cls_token"] = (1, 1, d)
    s["backbone.vision_embed.proj.weight"] = (d, 3, patch, patch)
    s["backbone.vision_embed.proj.bias"] = (d,)
    s["backbone.encoder.embed_positions.A.weight"] = ((res // patch) ** 2 + 1 + 2, d)
    s["backbone.encoder.embed_positions.B.weight"] = (max_src_pos, d)

    def enc(prefix, nl):
        for i in range(nl):
            b = prefix + f"layers.{i}."
            for br in "AB":
                for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
                    s[b + f"self_attn.{nm}.{br}.weight"] = (d, d)
                    s[b + f"self_attn.{nm}.{br}.bias"] = (d,)
                for ln in ("self_attn.inner_attn_ln", "self_attn_layer_norm", "final_layer_norm"):
                    s[b + f"{ln}.{br}.weight"] = (d,)
                    s[b + f"{ln}.{br}.bias"] = (d,)
                s[b + f"ffn.{br}.fc1.weight"] = (4 * d, d)
                s[b + f"ffn.{br}.fc1.bias"] = (4 * d,)
                s[b + f"ffn.{br}.fc2.weight"] = (d, 4 * d)
                s[b + f"ffn.{br}.fc2.bias"] = (d,)
                s[b + f"ffn.{br}.ffn_layernorm.weight"] = (4 * d,)
                s[b + f"ffn.{br}.ffn_layernorm.bias"] = (4 * d,)
        for br in "AB":
            s[prefix + f"layer_norm.{br}.weight"] = (d,)
            s[prefix + f"layer_norm.{br}.bias"] = (d,)

    enc("backbone.encoder.", layers)
    enc("backbone_vl.", vl_layers)
    for h in ("itc_text_proj", "itc_image_proj", "itc_vl_text_proj", "itc_vl_image_proj"):
        s[h + ".fc.weight"] = (out, d)
    return s
