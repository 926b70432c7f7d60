"""Full-DEPTH / real-WIDTH parity of the three BASELINE workloads against the CPU oracle (VERDICT r4 "missing" 2).

    python tests/real_width_case.py l14|b16|vtp8|vtp8t|dmae12 [cuda:0|cpu]        -> one JSON line, exit code 1 if a gate fails

The tiny-model fixtures (2 + 1 layers, d = 128) cannot see what accumulates over 24 bf16 layers at d = 1024 or over 197-token ViT-B/16 frames;
this runs the PRODUCT model at the bench's own dimensions on a few pairs and `oracle.step.*` (fp32, host) on the same name-keyed weights and
inputs.  Every number asserted on is also printed, so DESIGN.md quotes measured deviations, not gates.

  l14     VLMo `large`, patch 14: 21 + 3 layers, d = 1024, 16 heads, 257 + 77 tokens, vocab 115244 (bench.py M2_WORKLOADS["l14"]); 4 pairs, ragged
          captions; oracle.step.m2_itc.  Reference: prj/M2_Encoder/vlmo/modules/vlmo_module.py:323-405, torchscale/architecture/encoder.py:388-482.
  b16     VLMo `base`, patch 16: 9 + 3 layers, d = 768, 12 heads, 197 + 77 tokens (BASELINE configs[1]); 4 pairs; oracle.step.m2_itc.
  vtp8    prj/base_vtp `univl`, clip arch ViT-B/16 + BERT-base, 8 clips, stage1 + stage2 (bench.py VTP_WORKLOADS["vtp8"]); 2 videos;
          oracle.step.univl_stage1 + univl_stage2.  Reference: prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:357-443.
  vtp8t   config 3 with its temporal module: 8 frames, 77-token ragged captions, stage1 + stage3 = 4-layer seqTransf + WTI + CrossEn (bench.py VTP_WORKLOADS["vtp8t"]); 2 videos.
  dmae12  prj/dmae_vtp `univl`, 12 frames x 30 words, stage1 + stage3 with the 4-layer seqTransf header, WTI, NegNCE (TPM-CL off: its hinge set is
          pinned op by op, see tests/model_cases.py); 2 videos; oracle.step.univl_stage1 + dmae_stage3.  Reference:
          prj/dmae_vtp/roi_univl/univl/model/univl_video_ret.py:457-476, dmae_utils.py:186-278.

Runs as its own process: dmae_vtp's package shares its name with base_vtp's, and the l14 model's 40 GB leave with the process.
TEST INFRASTRUCTURE: imports oracle/.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ant-multi-modal-framework_amd")
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), PKG, ROOT]

import weightgen as W  # noqa: E402

CLIP_B16 = dict(width=768, layers=12, heads=12, patch=16, res=224, out_dim=768, vocab=21128, hidden=768, inter=3072, bert_layers=12, bert_heads=12, max_pos=512)


def cmp_grads(named, P, skip=()):
    """per-parameter (cosine, relative norm error, name, error relative to max(own norm, 1 % of the largest parameter-gradient norm), own norm / largest) of the
    product gradients against the oracle's; parameters whose true gradient vanishes (key-projection biases: softmax shift invariance) apart; and the cosine of the
    whole-model gradient (all parameters concatenated)."""
    rows, zero = [], []
    top = max(float(p.grad.norm()) for p in P.values() if p.grad is not None)
    dot = gg = rr = 0.0
    for n, p in P.items():
        if p.grad is None or n not in named or any(s in n for s in skip):
            continue
        got = named[n].grad
        assert got is not None, f"no gradient for {n}"
        got, ref = got.detach().float().flatten().cpu(), p.grad.flatten()
        gn, rn = float(got.norm()), float(ref.norm())
        dot, gg, rr = dot + float(torch.dot(got.double(), ref.double())), gg + float(torch.dot(got.double(), got.double())), rr + float(torch.dot(ref.double(), ref.double()))
        if rn < 1e-5 * top:
            zero.append((gn / top, n))
            continue
        rows.append((float(torch.dot(got, ref)) / max(gn * rn, 1e-30), abs(gn - rn) / rn, n, float((got - ref).norm()) / max(rn, 1e-2 * top), rn / top))
    GLOBAL["cos"] = dot / max((gg * rr) ** 0.5, 1e-30)
    return rows, zero


GLOBAL = {}


def copy_weights(model, P):
    named = dict(model.named_parameters())
    missing = [n for n in named if n not in P]
    with torch.no_grad():
        for n, p in named.items():
            if n in P:
                assert tuple(p.shape) == tuple(P[n].shape), (n, tuple(p.shape), tuple(P[n].shape))
                p.copy_(P[n])
    return missing


def ragged_mask(lengths, seq):
    return (torch.arange(seq)[None, :] < torch.tensor(lengths)[:, None]).long()


def report_rows(rows, pick=()):
    """min cosine / worst norm over ALL parameters (with the parameter's share of the largest gradient norm: a tiny, strongly cancelling gradient -- the last BERT layer's
    query / key weights see ONE query per sequence, the pooled [CLS] -- carries flash-attention's bf16 noise at a few 1e-4 of the largest norm whatever the kernels) and
    the gated figures: `worst_err` = max over parameters of |g - g_ref| / max(|g_ref|, 1 % of the largest norm), `global_cos` = cosine of the whole-model gradient."""
    rows = sorted(rows)
    big = [r for r in rows if r[4] >= 1e-2]
    out = dict(n=len(rows), global_cos=round(GLOBAL["cos"], 6), worst_err=max(rows, key=lambda r: r[3])[2:5], min_cos=rows[0][:3] + rows[0][4:], worst_norm=max(rows, key=lambda r: r[1])[:3],
               min_cos_above_1pct=(min(big)[:3] if big else None), n_above_1pct=len(big))
    for tag in pick:
        sel = [r for r in rows if tag in r[2]]
        if sel:
            out[tag] = dict(n=len(sel), min_cos=round(min(r[0] for r in sel), 5), worst_norm=round(max(r[1] for r in sel), 4), worst_err=round(max(r[3] for r in sel), 4))
    return out


def grad_gates(g, max_err=0.15, min_cos=0.99, min_global=0.999):
    """whole-model direction; every parameter within 15 % of max(its own norm, 1 % of the largest) -- cosine 0.99 is an error of 14 %; parameters carrying >= 1 % of the largest
    norm also by cosine >= 0.99.  (max_err / min_cos: the DMAE case, whose scores route gradients through arg-max selections, states its own.)"""
    bad = []
    if g["global_cos"] < min_global:
        bad.append("global gradient direction")
    if g["worst_err"][1] > max_err:
        bad.append("gradient error")
    if g["min_cos_above_1pct"] is not None and (g["min_cos_above_1pct"][0] < min_cos):
        bad.append("gradient direction of a large parameter")
    return bad


def case_l14(dev, which="l14"):
    sys.path.insert(0, os.path.join(PKG, "prj", "M2_Encoder"))
    from oracle import step as ostep
    from oracle.shapes import m2_shapes
    from vlmo.config import default_config
    from vlmo.modules.vlmo_module import VLMo

    full = os.environ.get("ANTMMF_REAL_WIDTH_SMALL") != "1"   # (the CPU emulator cannot run 24 layers at d = 1024 in test time: plumbing check only)
    m = dict(beit_version="large", encoder_embed_dim=1024, out_embed_dim=1024, encoder_layers=21, beit3_vl_layers=3, image_size=224, patch_size=14,
             vocab_size=115244, max_text_len=77)
    if which == "b16":   # BASELINE configs[1]: M2 `base`, patch 16, 9 + 3 layers, d = 768 (bench.py M2_WORKLOADS["b16"])
        m = dict(beit_version="base", encoder_embed_dim=768, out_embed_dim=768, encoder_layers=9, beit3_vl_layers=3, image_size=224, patch_size=16,
                 vocab_size=64010, max_text_len=77)
    if not full:
        m.update(encoder_embed_dim=128, out_embed_dim=128, encoder_layers=2, beit3_vl_layers=1, image_size=28, vocab_size=500, max_text_len=12, encoder_attention_heads=2)
    d, seq = m["encoder_embed_dim"], m["max_text_len"]
    heads = d // 64
    t0 = time.time()
    P = W.fill_dict(m2_shapes(d=d, layers=m["encoder_layers"], vl_layers=m["beit3_vl_layers"], patch=m["patch_size"], res=m["image_size"], vocab=m["vocab_size"], out=m["out_embed_dim"]))
    cfg = default_config()
    cfg.update(m)
    model = VLMo(cfg)
    missing = copy_weights(model, P)
    assert all(n.startswith(("norm.", "pooler.")) for n in missing), missing[:8]
    model = model.to(dev).train()
    B = 4
    mask = ragged_mask([seq, 30 if full else 7, 9, 55 if full else 11], seq)

    def batch(k):
        im = (W.data_tensor(f"fd.image.{k}", (B, 3, m["image_size"], m["image_size"])) * 0.25 + 0.5).clamp(0, 1)
        return im, W.data_ints(f"fd.ids.{k}", (B, seq), 1, m["vocab_size"]) * mask

    t_build = time.time() - t0
    for v in P.values():
        v.requires_grad_(True)
    img, ids = batch(0)
    out = model({"image": [img.to(dev)], "text_ids": ids.to(dev), "text_masks": mask.to(dev)})
    loss = out["losses"]["itc_loss"] + out["losses"]["itc_vl_loss"]
    t1 = time.time()
    ref = ostep.m2_itc(P, img, ids, mask, heads=heads, patch=m["patch_size"])
    ref["loss"].backward()
    t_oracle = time.time() - t1
    loss.backward()
    rel = [(float(loss.detach()) - float(ref["loss"].detach())) / abs(float(ref["loss"].detach()))]
    emb = {}
    with torch.no_grad():
        oi = model.infer_image({"image": [img.to(dev)]})
        ot = model.infer_text({"text_ids": ids.to(dev), "text_masks": mask.to(dev)})
        for k, got, want in (("img.cls", oi["cls_feats"], ref["img"]["cls_feats"]), ("txt.cls", ot["cls_feats"], ref["txt"]["cls_feats"]),
                             ("img.vl", oi["cls_vlffn_feats"], ref["img"]["cls_vlffn_feats"]), ("txt.vl", ot["cls_vlffn_feats"], ref["txt"]["cls_vlffn_feats"])):
            g, r = got.float().cpu(), want.detach()
            emb[k] = dict(max_abs=round(float((g - r).abs().max()), 5), min_row_cos=round(float(torch.nn.functional.cosine_similarity(g, r, dim=-1).min()), 6))
        lg = (model.logit_scale.exp() * oi["cls_feats"] @ ot["cls_feats"].t()).float().cpu()
        emb["logits_max_abs"] = round(float((lg - ref["logits"].detach()).abs().max()), 5)
        emb["logits_ref_absmax"] = round(float(ref["logits"].detach().abs().max()), 3)
        for k in range(1, 3):
            im_k, ids_k = batch(k)
            o_k = model({"image": [im_k.to(dev)], "text_ids": ids_k.to(dev), "text_masks": mask.to(dev)})
            r_k = float(ostep.m2_itc(P, im_k, ids_k, mask, heads=heads, patch=m["patch_size"])["loss"])
            rel.append((float(o_k["losses"]["itc_loss"] + o_k["losses"]["itc_vl_loss"]) - r_k) / abs(r_k))
    rows, zero = cmp_grads(dict(model.named_parameters()), P)
    depth = ("encoder.layers.0.", f"encoder.layers.{m['encoder_layers'] // 2}.", f"encoder.layers.{m['encoder_layers'] - 1}.", "backbone_vl.layers.0.",
             f"backbone_vl.layers.{m['beit3_vl_layers'] - 1}.", "text_embed", "vision_embed", "itc_")
    rep = dict(case=which, full=full, pairs=B, loss=float(loss), ref_loss=float(ref["loss"]), loss_rel=[round(r, 6) for r in rel], loss_rel_mean=round(sum(rel) / len(rel), 6),
               embeddings=emb, grads=report_rows(rows, depth), zero_grads=sorted(zero, reverse=True)[:2], seconds=dict(build=round(t_build, 1), oracle_fwd_bwd=round(t_oracle, 1)))
    gates = []
    # north_star: loss within 1e-3 relative -- on the mean over the seeded batches (a 4-pair InfoNCE turns the embeddings' bf16 error into a per-batch sigma of that order,
    # tests/model_cases.py::case_m2_itc_vs_oracle), 5e-3 on any single batch
    if abs(rep["loss_rel_mean"]) > 1e-3 or max(abs(r) for r in rel) > 5e-3:
        gates.append("loss")
    if min(e["min_row_cos"] for e in emb.values() if isinstance(e, dict)) < 0.999:
        gates.append("embedding direction")
    gates += grad_gates(rep["grads"])
    if any(z[0] > 1e-2 for z in zero):
        gates.append("zero gradients")
    return rep, gates


def build_univl(prj, model_cfg, P, dev):
    sys.path.insert(0, os.path.join(PKG, "prj", prj))
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    model = UnivlForVideoTextRetrieval(Configuration(model_cfg))
    missing = copy_weights(model, P)
    model = model.to(dev).train()
    if hasattr(model, "dropout"):
        model.dropout.p = 0.0
    return model, missing


def clip_cfg(c, **extra):
    enc = dict(image_encoder=dict(type="VitImageEncoder", params=dict(model_name="ViT-B-16", input_resolution=c["res"], patch_size=c["patch"], width=c["width"], layers=c["layers"],
                                                                      out_dim=c["out_dim"], pretrained=False)),
               text_encoder=dict(type="RobertBertEncoder", params=dict(pretrained=False, vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["inter"],
                                                                       num_hidden_layers=c["bert_layers"], num_attention_heads=c["bert_heads"], max_position_embeddings=c["max_pos"],
                                                                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, out_dim=c["out_dim"], is_proj=True)))
    return dict(training_head_type="video_text_retrieval", arch_type="clip", with_moco=False, hidden_size=c["hidden"], **enc, **extra)


def video_inputs(tag, B, n, c, seq, lengths, dev):
    frames = W.data_tensor(f"{tag}.frames", (B, n, 3, c["res"], c["res"]))
    mask = ragged_mask(lengths, seq)
    ids = W.data_ints(f"{tag}.ids", (B, seq), 1000, c["vocab"]) * mask
    ids[:, 0] = 101
    img_input = dict(image_data=frames.to(dev), image_pad_mask=torch.zeros(B, n, c["res"], c["res"], dtype=torch.bool, device=dev), image_n_clips=[n] * B, image_num_frames=[1] * B)
    cap_input = dict(caption_input_ids=ids.to(dev), caption_input_mask=mask.to(dev), caption_raw_input_ids=ids.to(dev))
    return frames, ids, mask, img_input, cap_input


def small_clip():
    return dict(CLIP_B16, width=128, layers=2, heads=2, patch=8, res=32, out_dim=128, vocab=2000, hidden=128, inter=512, bert_layers=2, bert_heads=2, max_pos=64)


def case_vtp8(dev):
    import tiny_models
    from oracle import step as ostep

    full = os.environ.get("ANTMMF_REAL_WIDTH_SMALL") != "1"
    c = CLIP_B16 if full else small_clip()
    n, seq, B = 8, 77 if full else 12, 2
    h = c["hidden"]
    shapes = tiny_models.clip_arch_shapes(c)
    shapes.update({"similarity_dense.0.weight": (2 * h, h), "similarity_dense.0.bias": (2 * h,), "similarity_dense.2.weight": (1, 2 * h), "similarity_dense.2.bias": (1,)})
    P = W.fill_dict(shapes)
    model, missing = build_univl("base_vtp", clip_cfg(c, training_stage="stage1+stage2", with_cross_encoder=True, cross_chunk_rows=B), P, dev)
    frames, ids, mask, img_input, cap_input = video_inputs("fd.vtp8", B, n, c, seq, [seq, 21 if full else 7], dev)
    for v in P.values():
        v.requires_grad_(True)
    out = model(img_input, cap_input)
    l1, l2 = out["losses"]["level1_similarity_loss"], out["losses"]["level2_similarity_loss"]
    t1 = time.time()
    r1 = ostep.univl_stage1(P, frames, ids, mask, n, c["heads"], c["patch"], c["bert_heads"])
    r2 = ostep.univl_stage2(P, frames, ids, mask, n, c["heads"], c["patch"], c["bert_heads"])
    # the gradient check runs on a scalar that does not cancel (tests/model_cases.py::case_univl_stage2): level-1 loss + fixed positive weights on the cross-encoder scores
    wpin = (W.data_tensor("fd.vtp8.pin", (B, B)).abs() + 0.5)
    (r1["loss"] + (r2["l2_simi"] * wpin).sum()).backward()
    t_oracle = time.time() - t1
    (l1 + (out["l2_simi"].float() * wpin.to(dev)).sum()).backward()
    rows, zero = cmp_grads(dict(model.named_parameters()), P)
    s2 = out["l2_simi"].detach().float().cpu()
    rep = dict(case="vtp8", full=full, videos=B, clips=n, loss1=float(l1), ref_loss1=float(r1["loss"]), loss1_rel=round((float(l1) - float(r1["loss"])) / abs(float(r1["loss"])), 6),
               loss2=float(l2), ref_loss2=float(r2["loss"]), loss2_rel=round((float(l2) - float(r2["loss"])) / abs(float(r2["loss"])), 6),
               l2_simi_max_abs=round(float((s2 - r2["l2_simi"].detach()).abs().max()), 5), l2_simi_ref_absmax=round(float(r2["l2_simi"].detach().abs().max()), 4),
               l1_simi_max_abs=round(float((out["l1_simi"].detach().float().cpu() - r1["l1_simi"].detach()).abs().max()), 6),
               grads=report_rows(rows, ("resblocks.0.", "resblocks.11.", "encoder.layer.0.", "encoder.layer.11.", "similarity_dense", "embeddings")),
               zero_grads=sorted(zero, reverse=True)[:2], missing=missing[:4], seconds=dict(oracle_fwd_bwd=round(t_oracle, 1)))
    gates = []
    if abs(rep["loss1_rel"]) > 1e-3:
        gates.append("loss1")
    if abs(rep["loss2_rel"]) > 2e-3:    # the gate of the tiny fixture (cross-encoder scores through 12 more bf16 layers + an MLP)
        gates.append("loss2")
    gates += grad_gates(rep["grads"])
    return rep, gates


# Gate of the HEAD's part of the level-3 loss deviation (product head vs the fp32 oracle head on the same, product, tower outputs).  Measured on MI355X
# (profiles/r6_dmae_level3_head_tower_split.txt): seqTransf heads +1.2 % (dmae12) / +2.4 % (vtp8t) -- four bf16 temporal layers in front of a logit scale of 100; either sign
# -- against a towers' part of -2.8 % / -3.0 %; the meanP head with TPM-CL on (no bf16 layer in the head): 1e-5-class, gated at 2e-3.
HEAD_GATE = {"dmae12tpm": 2e-3}


def case_dmae12(dev, which="dmae12"):
    import tiny_models
    from oracle import step as ostep

    full = os.environ.get("ANTMMF_REAL_WIDTH_SMALL") != "1"
    c = CLIP_B16 if full else small_clip()
    n, seq, B, L = 12, 30 if full else 12, 2, 4 if full else 2
    loss_type, lengths = "negNCE", None
    if which == "vtp8t":   # config 3 with its temporal module (bench.py VTP_WORKLOADS["vtp8t"]): 8 frames, 77-token ragged captions, CrossEn
        n, seq, loss_type = 8, 77 if full else 12, "cross_entropy"
        lengths = [seq, 21 if full else 7]
    # "dmae12tpm" (round 6): the DMAE step with TPM-CL ON at real width -- partial-order margin losses of type 4 (dmae_utils.py:280-463) on the meanP header (the header the
    # oracle's margin loss restates and ops_dmae_tpmcl.pt pins), 4 videos = one 4 x 4 caption x video block
    tpm = which == "dmae12tpm"
    sim_header = "meanP" if tpm else "seqTransf"
    if tpm:
        B = 4 if full else 2
    h = c["hidden"]
    # l3_with_nfc False: the second-best-frame term is a function of ARG-max indices (dmae_utils.py:105-118) -- discontinuous in the features, so two correct builds differ by
    # whole terms when a near-tie flips (measured at this width with it on: 5 % on the scores); it is pinned on the reference's own fixtures (ops_dmae_wti.pt, with and without)
    extra = dict(training_stage="stage1+stage3", with_cross_encoder=False, l3_interaction="wti", l3_with_nfc=False, l3_wti_arch=1, l3_sim_header=sim_header,
                 l3_sim_header_hidden_layer=L, l3_partial_type=4 if tpm else -1, l3_max_frames=n, l3_max_words=seq, l3_loss_type=loss_type)
    sys.path.insert(0, os.path.join(PKG, "prj", "dmae_vtp"))
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    if os.environ.get("ANTMMF_DMAE_BF16_STREAM") == "0":   # A/B of the temporal transformer's state precision (round 6): the opt-in fp32 stream instead of the fused bf16 layer
        from roi_univl.univl.model import dmae_utils as _du

        _du.TransformerClip.FP32_STREAM = True
    model = UnivlForVideoTextRetrieval(Configuration(clip_cfg(c, **extra)))
    # the oracle's parameter table = the product model's own names (towers as in tiny_models.clip_arch_shapes + the dmae_utils.* head), filled by name
    P = W.fill_dict({k: tuple(v.shape) for k, v in model.named_parameters()})
    assert set(tiny_models.clip_arch_shapes(c)) <= set(P) | {"module.text_encoder.pooler.dense.weight", "module.text_encoder.pooler.dense.bias"}
    copy_weights(model, P)
    model = model.to(dev).train()
    frames, ids, mask, img_input, cap_input = video_inputs("fd." + which, B, n, c, seq, lengths or [seq] * B, dev)   # (DMAE's predictors are built for exactly l3_max_words tokens)
    for v in P.values():
        v.requires_grad_(True)
    out = model(img_input, cap_input)
    l1, l3 = out["losses"]["level1_similarity_loss"], out["losses"]["level3_similarity_loss"]
    t1 = time.time()
    r1 = ostep.univl_stage1(P, frames, ids, mask, n, c["heads"], c["patch"], c["bert_heads"])
    r3 = ostep.dmae_stage3(P, frames, ids, mask, n, c["heads"], c["patch"], c["bert_heads"], loss_type=loss_type, with_va=False, sim_header=sim_header, sim_layers=L)
    from oracle import losses as olosses
    from oracle import towers as otowers

    Pd = otowers._sub(P, "dmae_utils.")
    r_margin = olosses.dmae_tpmcl_margin_loss(Pd, *r3["feats"][:3], mask.float(), r3["feats"][3], 4) if tpm else None
    # Which part of the level-3 deviation is the HEAD's and which the TOWERS' (round 6): the oracle's fp32 head -- and its fp32 TPM-CL margin loss -- evaluated
    # on the PRODUCT's tower outputs (word / frame token features and sentence embedding exactly as DmaeUtils.get_similarity_logits receives them).  head part = product loss -
    # that; tower part = that - the all-fp32 oracle.
    with torch.no_grad():
        cap_in, vis_in, _, _ = model.module.get_l2_input(img_input, cap_input)
        fp = lambda t: t.detach().float().cpu()   # noqa: E731
        hmix = ostep.dmae_stage3_head({k: v.detach() for k, v in P.items()}, fp(cap_in[0]), fp(vis_in[0]), fp(vis_in[1]), fp(cap_in[2]), mask, loss_type=loss_type, with_va=False,
                                      sim_header=sim_header, sim_layers=L)
        hmix_margin = float(olosses.dmae_tpmcl_margin_loss({k: v.detach() for k, v in Pd.items()}, *hmix["feats"][:3], mask.float(), hmix["feats"][3], 4)) if tpm else 0.0
    ref3 = float(r3["loss"]) + (float(r_margin) if tpm else 0.0)
    mix3 = float(hmix["loss"]) + hmix_margin
    # gradient check on a scalar that is alive whatever the scores are (with l3_with_nfc on, this random-weight model's scores are O(1000): NegNCE clamps its softmax at
    # 1e-6 and sits ON the clamp, zero gradient on both sides): the level-3 head is driven through fixed positive weights on the [T, V] token-wise scores
    wpin = (W.data_tensor("fd." + which + ".pin", (B, B)).abs() + 0.5)
    if tpm:   # the margin loss joins both scalars: its gradient reaches the TPM-CL predictors and, through the features, both towers
        (r1["loss"] + (r3["l3_simi"] * wpin).sum() / 100.0 + r_margin).backward()
        t_oracle = time.time() - t1
        (l1 + (out["l3_simi"].float() * wpin.to(dev)).sum() / 100.0 + (l3 - (model.loss_fct(out["l3_simi"]) + model.loss_fct(out["l3_simi"].t())) / 2)).backward()
    else:
        (r1["loss"] + (r3["l3_simi"] * wpin).sum() / 100.0).backward()
        t_oracle = time.time() - t1
        (l1 + (out["l3_simi"].float() * wpin.to(dev)).sum() / 100.0).backward()
    rows, zero = cmp_grads(dict(model.named_parameters()), P)
    s3 = out["l3_simi"].detach().float().cpu()
    rep = dict(case=which, full=full, videos=B, frames=n, loss1=float(l1), ref_loss1=float(r1["loss"]), loss1_rel=round((float(l1) - float(r1["loss"])) / abs(float(r1["loss"])), 6),
               loss3=float(l3), ref_loss3=ref3, loss3_rel=round((float(l3) - ref3) / abs(ref3), 6),
               loss3_head_part=round((float(l3) - mix3) / abs(ref3), 6), loss3_tower_part=round((mix3 - ref3) / abs(ref3), 6),
               l3_simi_head_part_max_abs=round(float((out["l3_simi"].detach().float().cpu() - hmix["l3_simi"]).abs().max()), 6),
               margin=dict(ref=float(r_margin), fp32_head_on_product_features=hmix_margin) if tpm else None,
               l3_simi_max_abs=round(float((s3 - r3["l3_simi"].detach()).abs().max()), 6), l3_simi_ref_absmax=round(float(r3["l3_simi"].detach().abs().max()), 4),
               grads=report_rows(rows, ("resblocks.0.", "resblocks.11.", "encoder.layer.0.", "encoder.layer.11.", "dmae_utils", "embeddings")),
               zero_grads=sorted(zero, reverse=True)[:2], seconds=dict(oracle_fwd_bwd=round(t_oracle, 1)))
    if tpm:
        rep["head_gradients_same_features"] = tpm_head_gradients(model, P, Pd, cap_in, vis_in, mask, olosses, ostep, loss_type, sim_header, L, dev)
    gates = []
    if abs(rep["loss1_rel"]) > 1e-3:
        gates.append("loss1")
    # (the level-3 LOSS of this 2-video batch is reported, not gated: logit scale 100 turns the 4 % score deviations discussed below into 1 - 2 % on the loss; the six-batch
    # contract test tests/model_cases.py::case_dmae_stage3_loss_contract and the reference fixture e2e_dmae_stage3.pt are what hold it)
    # WTI scores are sums of MAXIMA over tokens: a near-tie that bf16 noise flips changes a score by the gap between two candidates, and moves that score's whole gradient from
    # one token to another (measured on MI355X at this size: scores 4 % of their range, the patch-embedding gradient cosine 0.987, whole-model cosine 0.9992).  Gates: 8e-2 on the scores
    # (next comment); every parameter within 25 % of max(own norm, 1 % of the largest), large parameters cosine >= 0.98, whole model >= 0.998 (measured 0.99920 / 0.99995)
    # (scores: the ORACLE under torch's bf16 autocast is itself 2.1 % (vtp8t) / 3.0 % (dmae12) of the score range away from its fp32 self on these batches; this build, whose
    # residual stream through the 4 temporal layers is bf16 as well, 5.4 % / 4.0 %: gate 8 %)
    if rep["l3_simi_max_abs"] > 8e-2 * rep["l3_simi_ref_absmax"]:
        gates.append("l3_simi")
    # round 6: the level-3 loss IS gated now, on the part that is this build's head (given the same tower outputs, the product's head against the fp32 oracle head); the
    # towers' part (bf16 ViT / BERT features under a logit scale of 100) is reported next to it.  Gate sizes: measured values in profiles/r6_real_width.jsonl
    if abs(rep["loss3_head_part"]) > HEAD_GATE.get(which, 5e-2) or rep["l3_simi_head_part_max_abs"] > 4e-2 * rep["l3_simi_ref_absmax"]:
        gates.append("loss3_head_part")
    if tpm:
        # TPM-CL's margin losses are DIFFERENCES of a full and an importance-masked score (anchor - partial, dmae_utils.py:379-388) behind a discrete token selection: their gradient
        # is a cancellation that amplifies the towers' bf16 noise the further upstream a parameter sits (measured: the patch embedding at cosine 0.85, whole model 0.9994) -- like the
        # stage-2 loss gradient (tests/model_cases.py::case_univl_stage2).  End to end only the whole-model direction is gated; the HEAD's gradients are gated where they can be
        # compared exactly: product head against oracle head on the SAME (product) features, with respect to those features and to every head parameter.
        if rep["grads"]["global_cos"] < 0.999:
            gates.append("global gradient direction")
        hg = rep["head_gradients_same_features"]
        if hg["min_cos"] < 0.999 or hg["max_rel_err"] > 2e-2 or abs(hg["margin_rel"]) > 1e-4:
            gates.append("head gradients on the same features")
    else:
        gates += grad_gates(rep["grads"], max_err=0.25, min_cos=0.98, min_global=0.998)
    return rep, gates


def tpm_head_gradients(model, P, Pd, cap_in, vis_in, mask, olosses, ostep, loss_type, sim_header, L, dev):
    """Level-3 loss + TPM-CL margin loss of the PRODUCT head and of the ORACLE head on the same token features (the product's tower outputs as leaves): values and gradients
    with respect to the three feature tensors and every head parameter.  No tower is involved on either side, so this is the head's kernels against the head's arithmetic."""
    fp = lambda t: t.detach().float().cpu()   # noqa: E731
    du = model.dmae_utils
    for p_ in du.parameters():
        p_.grad = None
    leaves_p = [cap_in[0].detach().float().requires_grad_(True), vis_in[0].detach().float().requires_grad_(True), cap_in[2].detach().float().requires_grad_(True)]
    simi_p, margin_p = du.get_similarity_logits((leaves_p[1], vis_in[1]), (leaves_p[0], mask.to(dev), leaves_p[2], None, None), shaped=True, loose_type=True)
    loss_p = (model.loss_fct(simi_p) + model.loss_fct(simi_p.t())) / 2 + margin_p
    loss_p.backward()
    leaves_o = [fp(t).requires_grad_(True) for t in leaves_p]
    Pd_o = {k: v.detach().clone().requires_grad_(True) for k, v in Pd.items()}
    P_o = {("dmae_utils." + k): v for k, v in Pd_o.items()}
    ho = ostep.dmae_stage3_head(P_o, leaves_o[0], leaves_o[1], fp(vis_in[1]), leaves_o[2], mask, loss_type=loss_type, with_va=False, sim_header=sim_header, sim_layers=L)
    margin_o = olosses.dmae_tpmcl_margin_loss(Pd_o, *ho["feats"][:3], mask.float(), ho["feats"][3], 4)
    (ho["loss"] + margin_o).backward()
    rows = []
    for name, a, b in [("d/d word features", leaves_p[0].grad, leaves_o[0].grad), ("d/d frame features", leaves_p[1].grad, leaves_o[1].grad), ("d/d sentence feature", leaves_p[2].grad, leaves_o[2].grad)] + \
                      [(k, dict(du.named_parameters())[k].grad, v.grad) for k, v in Pd_o.items() if k in dict(du.named_parameters())]:
        if a is None or b is None or float(b.norm()) == 0.0:
            continue
        a, b = a.detach().float().cpu().flatten(), b.detach().float().flatten()
        rows.append([name, float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)), float((a - b).norm()), float(b.norm())])
    big = max(r[3] for r in rows)
    # error relative to max(own norm, 1e-3 of the largest): the bias of a Linear(D, 1) in front of a softmax has an exactly zero gradient (shift invariance) -- both sides hold round-off there
    rows = [(r[0], r[1] if r[3] > 1e-3 * big else 1.0, r[2] / max(r[3], 1e-3 * big)) for r in rows]
    return dict(n=len(rows), min_cos=min(r[1] for r in rows), max_rel_err=max(r[2] for r in rows), worst=max(rows, key=lambda r: r[2]),
                loss_rel=float((loss_p.detach().cpu() - (ho["loss"] + margin_o).detach()) / (ho["loss"] + margin_o).detach().abs()),
                margin_rel=float((margin_p.detach().cpu() - margin_o.detach()) / margin_o.detach().abs()))


def main():
    case = sys.argv[1]
    dev = torch.device(sys.argv[2] if len(sys.argv) > 2 else "cuda:0")
    torch.manual_seed(0)
    fn = dict(l14=case_l14, b16=lambda d: case_l14(d, "b16"), vtp8=case_vtp8, vtp8t=lambda d: case_dmae12(d, "vtp8t"), dmae12=case_dmae12, dmae12tpm=lambda d: case_dmae12(d, "dmae12tpm"))[case]
    rep, gates = fn(dev)
    rep["failed_gates"] = gates
    print("REALWIDTH " + json.dumps(rep, default=lambda o: list(o) if isinstance(o, tuple) else str(o)))
    sys.exit(1 if gates and os.environ.get("ANTMMF_REAL_WIDTH_REPORT_ONLY") != "1" else 0)


if __name__ == "__main__":
    main()
