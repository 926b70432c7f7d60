"""Pin the CPU oracle against fixtures produced by executing the reference (tests/golden/make_golden.py)."""
import torch

import weightgen as W
from oracle import losses, ops, step, towers

RTOL, ATOL = 2e-5, 2e-6


def close(a, b, rtol=RTOL, atol=ATOL):
    """fp32 parity: rtol elementwise, atol scaled by the tensor's own magnitude (sum-order noise)."""
    scale = max(1.0, float(b.detach().abs().max())) if b.numel() else 1.0
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol * scale)


def params_like(fix, prefix, salt=0):
    """Re-create the weights make_golden.py used: name-keyed fill, shapes taken from the stored grads."""
    out = {}
    for k, v in fix.items():
        if k.startswith(prefix):
            name = k[len(prefix):]
            out[name] = W.tensor_for(name, v.shape, salt).requires_grad_(True)
    return out


def test_clip_block(golden):
    g = golden("ops_clip_block.pt")
    P = params_like(g, "grad.")
    x = g["x"].clone().requires_grad_(True)
    y = towers.clip_block(P, x.transpose(0, 1), heads=2).transpose(0, 1)  # fixture is LND
    close(y, g["y"])
    (y * g["w"]).sum().backward()
    close(x.grad, g["dx"], 1e-4, 1e-5)
    for n, p in P.items():
        close(p.grad, g["grad." + n], 1e-4, 1e-5)


def test_bert_layer(golden):
    g = golden("ops_bert_layer.pt")
    P = params_like(g, "grad.")
    x = g["x"].clone().requires_grad_(True)
    y = towers.bert_layer(P, x, towers.bert_key_bias(g["mask"]), heads=2)
    close(y, g["y"])
    (y * g["w"]).sum().backward()
    close(x.grad, g["dx"], 1e-4, 1e-5)
    for n, p in P.items():
        close(p.grad, g["grad." + n], 1e-4, 1e-5)


def test_m2_layer(golden):
    g = golden("ops_m2_layer.pt")
    for br, pad in (("A", None), ("B", g["pad"])):
        P = params_like(g, "A.grad.")  # A.grad.* lists every parameter that got a grad in the A pass
        P.update(params_like(g, "B.grad."))
        x = g[f"{br}.x"].clone().requires_grad_(True)
        y = towers.m2_layer(P, x, br, heads=2, pad=pad)
        close(y, g[f"{br}.y"])
        (y * g[f"{br}.w"]).sum().backward()
        close(x.grad, g[f"{br}.dx"], 1e-4, 1e-5)
        for n, p in P.items():
            if f"{br}.grad.{n}" in g and p.grad is not None:
                close(p.grad, g[f"{br}.grad.{n}"], 1e-4, 1e-5)


def test_mil_nce(golden):
    g = golden("loss_mil_nce.pt")
    for key in sorted({k.split(".")[0] for k in g}):
        b, n = int(key[1:key.index("n")]), int(key[key.index("n") + 1:])
        s = g[f"{key}.sim"].clone().requires_grad_(True)
        loss = losses.mil_nce(s, b, n, g.get(f"{key}.weight"))
        close(loss, g[f"{key}.loss"])
        loss.backward()
        close(s.grad, g[f"{key}.dsim"], 1e-4, 1e-6)


def test_misc_losses(golden):
    g = golden("loss_misc.pt")
    pos, neg = g["moco.pos"].clone().requires_grad_(True), g["moco.neg"].clone().requires_grad_(True)
    loss = losses.moco(pos, neg, 0.05)
    close(loss, g["moco.loss"])
    loss.backward()
    close(pos.grad, g["moco.dpos"], 1e-4, 1e-6)
    close(neg.grad, g["moco.dneg"], 1e-4, 1e-6)
    for name, fn in (("crossen", losses.cross_en), ("negnce", losses.neg_nce)):
        s = g[f"{name}.sim"].clone().requires_grad_(True)
        loss = fn(s)
        close(loss, g[f"{name}.loss"])
        loss.backward()
        close(s.grad, g[f"{name}.dsim"], 1e-4, 1e-6)


def tiny_clip_params(g, tag):
    shapes = {}
    P = {}
    for k, v in g.items():
        if k.startswith(f"{tag}.gnorm."):
            P[k[len(f"{tag}.gnorm."):]] = None
    return P


def test_e2e_clip_arch(golden):
    import tiny_models

    g = golden("e2e_clip_arch.pt")
    for tag, n_clips in (("b4n1", 1), ("b3n2", 2)):
        P = tiny_models.clip_arch_params(requires_grad=True)
        out = step.univl_stage1(P, g[f"{tag}.image_data"], g[f"{tag}.input_ids"], g[f"{tag}.input_mask"],
                                n_clips, vit_heads=2, patch=8, bert_heads=2)
        close(out["text_embed"], g[f"{tag}.text_embed"], 1e-4, 1e-6)
        close(out["video_embed"], g[f"{tag}.video_embed"], 1e-4, 1e-6)
        close(out["l1_simi"], g[f"{tag}.l1_simi"], 1e-4, 1e-6)
        close(out["loss"], g[f"{tag}.loss"], 1e-5, 1e-6)
        out["loss"].backward()
        checked = 0
        for n, p in P.items():
            if f"{tag}.gnorm.{n}" in g:
                close(p.grad.norm(), g[f"{tag}.gnorm.{n}"], 2e-3, 1e-7)
                checked += 1
            if f"{tag}.grad.{n}" in g:
                close(p.grad, g[f"{tag}.grad.{n}"], 2e-3, 1e-7)
        assert checked > 50


def test_dmae_seqtransf(golden):
    """DmaeUtils._agg_visual_feat(sim_header="seqTransf"): frame position embedding + 2 masked CLIP blocks + residual."""
    import weightgen as W

    g = golden("ops_dmae_seqtransf.pt")
    names = [k[len("grad."):] for k in g if k.startswith("grad.")] + [k[len("gnorm."):] for k in g if k.startswith("gnorm.")]
    shapes = {"frame_position_embeddings.weight": (77, 128), "text_weight_fc.weight": (1, 128), "text_weight_fc.bias": (1,),
              "video_weight_fc.weight": (1, 128), "video_weight_fc.bias": (1,)}
    for i in range(2):
        b = f"transformerClip.resblocks.{i}."
        shapes.update({b + "attn.in_proj_weight": (384, 128), b + "attn.in_proj_bias": (384,), b + "attn.out_proj.weight": (128, 128),
                       b + "attn.out_proj.bias": (128,), b + "ln_1.weight": (128,), b + "ln_1.bias": (128,), b + "ln_2.weight": (128,),
                       b + "ln_2.bias": (128,), b + "mlp.c_fc.weight": (512, 128), b + "mlp.c_fc.bias": (512,),
                       b + "mlp.c_proj.weight": (128, 512), b + "mlp.c_proj.bias": (128,)})
    P = W.fill_dict(shapes)
    for v in P.values():
        v.requires_grad_(True)
    x = g["visual"].clone().requires_grad_(True)
    out, tok_mask, _ = towers.dmae_agg_visual_feat(P, x, g["mask"], heads=2, layers=2)
    close(out, g["out"], 1e-4, 1e-5)
    (out * g["w"]).sum().backward()
    close(x.grad, g["dvisual"], 1e-3, 1e-5)
    for n in names:
        if "grad." + n in g:
            close(P[n].grad, g["grad." + n], 2e-3, 1e-5)
        else:
            close(P[n].grad.norm(), g["gnorm." + n], 2e-3, 1e-6)


def test_e2e_clip_stage2(golden):
    """stage1 + stage2 (cross-encoder scores of every caption/video pair, MIL-NCE on the score matrix) vs the reference run."""
    import tiny_models

    g = golden("e2e_clip_stage2.pt")
    P = tiny_models.clip_arch_params(requires_grad=True, stage2=True)
    args = (g["s2.image_data"], g["s2.input_ids"], g["s2.input_mask"], 2)
    o1 = step.univl_stage1(P, *args, vit_heads=2, patch=8, bert_heads=2)
    o2 = step.univl_stage2(P, *args, vit_heads=2, patch=8, bert_heads=2)
    close(o1["loss"], g["s2.plain.loss1"], 1e-5, 1e-6)
    close(o2["l2_simi"], g["s2.plain.l2_simi"], 1e-4, 1e-5)
    close(o2["loss"], g["s2.plain.loss2"], 1e-5, 1e-6)
    (o1["loss"] + o2["loss"]).backward()
    checked = 0
    for n, p in P.items():
        if f"s2.plain.gnorm.{n}" in g:
            close(p.grad.norm(), g[f"s2.plain.gnorm.{n}"], 2e-3, 1e-7)
            checked += 1
    assert checked > 55


def test_dmae_wti(golden):
    """DmaeUtils.wti_interaction: wti / att_wti, with and without the second-best-frame term, forward and all gradients."""
    import weightgen as W

    g = golden("ops_dmae_wti.pt")
    shapes = {"text_weight_fc.weight": (1, 128), "text_weight_fc.bias": (1,), "video_weight_fc.weight": (1, 128), "video_weight_fc.bias": (1,)}
    for inter in ("wti", "att_wti"):
        for va in (True, False):
            tag = f"{inter}.va{int(va)}"
            P = W.fill_dict(shapes)
            for v in P.values():
                v.requires_grad_(True)
            t, w_, v = (g[k].clone().requires_grad_(True) for k in ("text", "word", "video"))
            out = losses.dmae_wti_interaction(P, t, w_, v, g["word_mask"], g["video_mask"], inter, va)
            close(out, g[f"{tag}.out"], 1e-5, 1e-6)
            (out * g["g"]).sum().backward()
            close(t.grad, g[f"{tag}.dtext"], 1e-4, 1e-6)
            close(v.grad, g[f"{tag}.dvideo"], 1e-4, 1e-6)
            if inter == "att_wti":
                close(w_.grad, g[f"{tag}.dword"], 1e-4, 1e-6)
            for n in shapes:
                if f"{tag}.grad.{n}" in g:
                    close(P[n].grad, g[f"{tag}.grad.{n}"], 1e-4, 1e-6)


def test_e2e_dmae_stage3(golden):
    """dmae_vtp stage1 + stage3 (WTI scores, NegNCE / CrossEn both directions) vs the reference run."""
    import tiny_models

    g = golden("e2e_dmae_stage3.pt")
    for loss_type in ("negNCE", "cross_entropy"):
        P = tiny_models.clip_arch_params(requires_grad=True, dmae=True)
        args = (g["s3.image_data"], g["s3.input_ids"], g["s3.input_mask"], 4)
        o1 = step.univl_stage1(P, *args, vit_heads=2, patch=8, bert_heads=2)
        o3 = step.dmae_stage3(P, *args, vit_heads=2, patch=8, bert_heads=2, loss_type=loss_type)
        close(o1["loss"], g[f"s3.{loss_type}.loss1"], 1e-5, 1e-6)
        close(o3["l3_simi"], g[f"s3.{loss_type}.l3_simi"], 1e-4, 1e-6)
        close(o3["loss"], g[f"s3.{loss_type}.loss3"], 1e-4, 1e-5)
        (o1["loss"] + o3["loss"]).backward()
        checked = 0
        for n, p in P.items():
            if f"s3.{loss_type}.gnorm.{n}" in g and p.grad is not None:
                close(p.grad.norm(), g[f"s3.{loss_type}.gnorm.{n}"], 3e-3, 1e-6)
                checked += 1
        assert checked > 55


def test_dmae_tpmcl(golden):
    """DmaeUtils.get_partial_similarity (TPM-CL margin losses, partial types 2 / 3 / 4) vs the reference run, incl. gradients."""
    import weightgen as W

    g = golden("ops_dmae_tpmcl.pt")
    D, Nw, V = 128, 12, 5
    shapes = {"text_weight_fc.weight": (1, D), "text_weight_fc.bias": (1,), "video_weight_fc.weight": (1, D), "video_weight_fc.bias": (1,)}
    for nm, F_, T_ in (("t2v_linear_xwp", 1, V), ("v2t_linear_xwp", V, Nw)):
        shapes.update({f"{nm}.q_proj.weight": (D, D), f"{nm}.k_proj.weight": (D, D), f"{nm}.qk_proj.weight": (T_, F_),
                       f"{nm}.attn_proj.0.weight": (T_, 2 * D), f"{nm}.attn_proj.0.bias": (T_, 2 * D),
                       f"{nm}.attn_proj.1.weight": (D // 2, 2 * D), f"{nm}.attn_proj.3.weight": (1, D // 2)})
    for ptype in (2, 3, 4):
        P = W.fill_dict(shapes)
        for v in P.values():
            v.requires_grad_(True)
        t, w_, v = (g[k].clone().requires_grad_(True) for k in ("text", "word", "video"))
        loss = losses.dmae_tpmcl_margin_loss(P, t, w_, v, g["word_mask"], g["video_mask"], ptype, cis_thresh=0.6)
        close(loss, g[f"p{ptype}.loss"], 1e-4, 1e-6)
        loss.backward()
        for nm, x in (("dtext", t), ("dword", w_), ("dvideo", v)):
            gr = x.grad if x.grad is not None else torch.zeros_like(x)
            close(gr.norm(), g[f"p{ptype}.{nm}.norm"], 2e-3, 1e-6)
            close(gr.flatten()[:256], g[f"p{ptype}.{nm}.probe"], 2e-3, 1e-6)
        for k in g:
            if k.startswith(f"p{ptype}.gnorm."):
                n = k[len(f"p{ptype}.gnorm."):]
                gn = P[n].grad.norm() if P[n].grad is not None else torch.tensor(0.0)
                close(gn, g[k], 3e-3, 1e-6)


def moco_queue(name, dim, K):
    import weightgen as W

    return torch.nn.functional.normalize(W.data_tensor(name, (dim, K)), dim=0)


def test_e2e_clip_moco(golden):
    """Two MoCo training steps (momentum key towers, queues, two-direction MoCo loss, enqueue) against the reference run
    (make_golden.py gen_e2e_clip_moco: K=64, M=0.5, an x1.05 weight perturbation standing in for the optimizer step)."""
    import tiny_models

    g = golden("e2e_clip_moco.pt")
    tag, n_clips = "moco", 2
    P = tiny_models.clip_arch_params(requires_grad=True)
    Pk = {k: v.detach().clone() for k, v in P.items()}
    queues = dict(txt=moco_queue("moco.txt_queue", 128, 64), img=moco_queue("moco.img_queue", 128, 16384), txt_ptr=0, img_ptr=0)
    for stp in (1, 2):
        for p in P.values():
            p.grad = None
        out = step.univl_stage1_moco(P, Pk, queues, g[f"{tag}.image_data"], g[f"{tag}.input_ids"], g[f"{tag}.input_mask"], n_clips,
                                     vit_heads=2, patch=8, bert_heads=2, momentum=0.5, temperature=0.05)
        close(out["loss"], g[f"{tag}.loss{stp}"], 1e-5, 1e-6)
        close(out["l1_simi"], g[f"{tag}.l1_simi{stp}"], 1e-4, 1e-6)
        out["loss"].backward()
        checked = 0
        for n, p in P.items():
            if f"{tag}.gnorm{stp}.{n}" in g:
                close(p.grad.norm(), g[f"{tag}.gnorm{stp}.{n}"], 2e-3, 1e-7)
                checked += 1
        assert checked > 50
        close(queues["txt"][:, :12], g[f"{tag}.txt_queue_head{stp}"], 1e-4, 1e-6)
        close(queues["img"][:, :20], g[f"{tag}.img_queue_head{stp}"], 1e-4, 1e-6)
        assert queues["txt_ptr"] == int(g[f"{tag}.txt_ptr{stp}"]) and queues["img_ptr"] == int(g[f"{tag}.img_ptr{stp}"])
        if stp == 1:
            with torch.no_grad():
                for p in P.values():
                    p.mul_(1.05)
    close(Pk["module.text_encoder.encoder.layer.0.attention.self.query.weight"][:4, :8], g[f"{tag}.key_probe"], 1e-5, 1e-7)


def test_e2e_m2(golden):
    import tiny_models

    g = golden("e2e_m2.pt")
    P = tiny_models.m2_params(g["param_names"], requires_grad=True)
    oi = towers.m2_infer_image(P, g["image"], heads=2, patch=8)
    ot = towers.m2_infer_text(P, g["text_ids"], g["text_masks"], heads=2)
    close(oi["image_feats"], g["img.image_feats"], 1e-4, 1e-5)
    for k in ("cls_feats", "cls_vlffn_feats"):
        close(oi[k], g[f"img.{k}"], 1e-4, 1e-6)
        close(ot[k], g[f"txt.{k}"], 1e-4, 1e-6)
    logits = P["logit_scale"].exp() * oi["cls_feats"] @ ot["cls_feats"].t()
    logits_vl = P["logit_vl_scale"].exp() * oi["cls_vlffn_feats"] @ ot["cls_vlffn_feats"].t()
    close(logits, g["logits"], 1e-4, 1e-5)
    pin = (logits * W.data_tensor("m2.wl", (3, 3))).sum() + (logits_vl * W.data_tensor("m2.wvl", (3, 3))).sum()
    close(pin, g["pin"], 1e-4, 1e-5)
    pin.backward()
    checked = 0
    for n, p in P.items():
        if f"gnorm.{n}" in g:
            close(p.grad.norm(), g[f"gnorm.{n}"], 2e-3, 1e-6)
            checked += 1
        if f"grad.{n}" in g:
            close(p.grad, g[f"grad.{n}"], 2e-3, 1e-6)
    assert checked > 50


def test_gather_scaling(golden):
    """W-rank gather: loss identical on every rank, local grad == W x single-process grad slice
    (antmmf/utils/distributed_utils.py:98-116)."""
    g = golden("gather_w2.pt")
    world = g["world"]
    t = [W.data_tensor("gather.t", (world * 3, 8))[r * 3:(r + 1) * 3].clone().requires_grad_(True) for r in range(world)]
    v = [W.data_tensor("gather.v", (world * 3, 8))[r * 3:(r + 1) * 3].clone().requires_grad_(True) for r in range(world)]
    gt = step.GatherWithGrad.apply(world, *t)
    gv = step.GatherWithGrad.apply(world, *v)
    loss = losses.mil_nce(gt @ gv.t(), world * 3, 1)
    loss.backward()
    for r in range(world):
        close(loss, g[f"rank{r}.loss"])
        close(t[r].grad, g[f"rank{r}.dt"], 1e-4, 1e-6)
        close(v[r].grad, g[f"rank{r}.dv"], 1e-4, 1e-6)
