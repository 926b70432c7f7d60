"""Backend-agnostic end-to-end parity cases for the product models (tiny dims), against the golden fixtures
produced by executing the reference (tests/golden/make_golden.py) and against the CPU oracle."""
import math
import os
import sys

import torch

import tiny_models
import weightgen as W
from kernel_cases import check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRJ = os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "base_vtp")
if PRJ not in sys.path:
    sys.path.insert(0, PRJ)

TINY_CLIP_CFG = dict(
    training_head_type="video_text_retrieval", arch_type="clip", training_stage="stage1", with_moco=False,
    with_cross_encoder=False, hidden_size=128,
    image_encoder=dict(type="VitImageEncoder", params=dict(
        model_name="ViT-tiny", input_resolution=32, patch_size=8, width=128, layers=2, out_dim=128, pretrained=False)),
    text_encoder=dict(type="RobertBertEncoder", params=dict(
        pretrained=False, vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
        num_attention_heads=2, max_position_embeddings=40, hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0, out_dim=128, is_proj=True)),
)



def grad_direction_report(named_params, g, prefix, zero_rel=1e-4):
    """Per-parameter gradient parity against the reference's FULL gradients (`<prefix>gfull.<name>`, bf16 storage): cosine of the
    flattened gradients and relative difference of their norms.  Parameters whose TRUE gradient vanishes (reference norm below
    `zero_rel` x the largest, e.g. the key-projection bias by softmax shift invariance) must stay small instead.
    Returns rows [cos, norm_rel, name] sorted worst-cosine first."""
    rows, refs = [], {}
    for n, p in named_params:
        k = f"{prefix}gfull.{n}"
        if k in g and p.grad is not None:
            refs[n] = (p.grad.detach().float().flatten().cpu(), g[k].float().flatten())
    top = max(float(r.norm()) for _, r in refs.values())
    for n, (got, ref) in refs.items():
        rn, gn = float(ref.norm()), float(got.norm())
        if rn < zero_rel * top:
            assert gn < 10 * zero_rel * top, f"{n}: gradient should vanish, got norm {gn}"
            continue
        rows.append([float(torch.dot(got, ref)) / max(gn * rn, 1e-30), abs(gn - rn) / rn, n])
    rows.sort()
    return rows


def assert_grad_directions(named_params, g, prefix, min_cos=0.995, max_norm_rel=0.05, min_checked=20):
    """north_star: "logits, loss and grads match the reference": every parameter's gradient points the reference's way (cosine) and
    has its length (norm), not only the length (a norm check cannot see a permuted head or a wrong sign)."""
    rows = grad_direction_report(named_params, g, prefix)
    assert len(rows) >= min_checked, len(rows)
    assert rows[0][0] >= min_cos, f"gradient direction off: {rows[:5]}"
    worst_norm = max(rows, key=lambda r: r[1])
    assert worst_norm[1] <= max_norm_rel, f"gradient norm off: {worst_norm}"
    return dict(min_cos=rows[0][:1] + rows[0][2:], worst_norm=worst_norm[1:], n=len(rows))

def build_tiny_univl(dev):
    import roi_univl  # noqa: F401  (registers encoders + model)
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    model = UnivlForVideoTextRetrieval(Configuration(TINY_CLIP_CFG))
    W.fill_module_(model)
    return model.to(dev).train()


def case_univl_stage1(dev, golden, tag="b4n1", n_clips=1, rtol=5e-2):
    """Product UnivlForVideoTextRetrieval (bf16 HIP path) vs the reference's outputs on the same weights / batch.
    Tolerance: bf16 activations through 2+2 layers; loss within 1e-3 relative (the contract of BASELINE.json's north_star); every
    parameter's gradient: cosine >= 0.995 against the reference's full gradient and norm within 5 %."""
    g = golden("e2e_clip_arch.pt")
    model = build_tiny_univl(dev)
    img = g[f"{tag}.image_data"].to(dev)
    ids, mask = g[f"{tag}.input_ids"].to(dev), g[f"{tag}.input_mask"].to(dev)
    bsz = img.shape[0]
    img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                     image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
    cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
    out = model(img_input, cap_input)
    loss = out["losses"]["level1_similarity_loss"]
    ref_loss = float(g[f"{tag}.loss"])
    assert abs(float(loss) - ref_loss) <= 1e-3 * abs(ref_loss), (float(loss), ref_loss)   # north_star: loss within 1e-3 rel
    check(f"{tag}.l1_simi", out["l1_simi"], g[f"{tag}.l1_simi"], rtol, 5e-2)
    cap, vis, _, _ = model.module.get_l2_input(img_input, cap_input)
    check(f"{tag}.text_embed", cap[2], g[f"{tag}.text_embed"], rtol, 3e-2)
    check(f"{tag}.video_embed", vis[2], g[f"{tag}.video_embed"], rtol, 3e-2)
    loss.backward()
    n_checked = 0
    worst = []
    for n, p in model.named_parameters():
        key = f"{tag}.gnorm.{n}"
        if key not in g:
            continue
        assert p.grad is not None, f"no grad for {n}"
        ref = float(g[key])
        got = float(p.grad.float().norm())
        worst.append([abs(got - ref), n, got, ref])
        n_checked += 1
        fk = f"{tag}.grad.{n}"
        if fk in g:
            check(fk, p.grad, g[fk], 1e-1, 1e-1)
    # relative error of each parameter's gradient norm.  Parameters whose TRUE gradient is zero (the key-projection
    # bias, by softmax shift invariance: the reference holds 1e-10 rounding noise there) are checked absolutely.
    top = max(w[3] for w in worst)
    kept = []
    for w in worst:
        if w[3] < 1e-6 * top:
            assert w[2] < 1e-3 * top, f"{w[1]}: gradient should vanish, got norm {w[2]}"
            continue
        w[0] = w[0] / w[3]
        kept.append(w)
    worst = sorted(kept, reverse=True)
    assert n_checked > 50
    assert worst[0][0] < 0.05, f"gradient norms off: {worst[:5]}"
    dirs = assert_grad_directions(model.named_parameters(), g, f"{tag}.", min_checked=50)
    return dict(loss=float(loss), ref_loss=ref_loss, worst_gnorm=worst[:3], directions=dirs)


def case_univl_registry(dev, golden):
    """the registry model `univl` built by build_model from a config, batch keys routed by prefix (group_inputs,
    univl_model.py:36-51), loss equal to the reference's on the golden batch, and get_optimizer_parameters' four groups
    {towers, new} x {decay, no decay} with the encoder lr decay (univl_video_ret.py:482-542) feeding the fused AdamW: after one step every
    parameter has moved by (about) ITS group's learning rate -- AdamW's first update is lr * sign-like, so the group lr is visible per element."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from antmmf.models.build import build_model
    from antmmf.optimizer import build_optimizer
    from antmmf.structures.sample import SampleList

    mcfg = Configuration(dict(TINY_CLIP_CFG, model="univl", encoder_lr_decay=0.1))
    model = build_model(mcfg)
    W.fill_module_(model.model)
    model = model.to(dev).train()
    g = golden("e2e_clip_arch.pt")
    img, ids, mask = g["b4n1.image_data"].to(dev), g["b4n1.input_ids"].to(dev), g["b4n1.input_mask"].to(dev)
    sl = SampleList(image_data=img, image_pad_mask=torch.zeros(4, img.shape[1], 32, 32, dtype=torch.bool, device=dev), image_n_clips=[1] * 4,
                    image_num_frames=[1] * 4, caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids, dataset_type="train")
    groups = model.group_inputs(sl)
    assert set(groups["image"]) == {"image_data", "image_pad_mask", "image_n_clips", "image_num_frames"} and len(groups["caption"]) == 3
    cfg = Configuration({"optimizer_attributes": {"type": "AdamW", "params": {"lr": 1e-3, "weight_decay": 0.0}}})
    pg = model.get_optimizer_parameters(cfg)
    assert [round(g_.get("lr", 1e-3), 8) for g_ in pg] == [1e-4, 1e-3, 1e-4, 1e-3] and [g_["weight_decay"] for g_ in pg] == [0.0, 0.0, 0.0, 0.0]
    cfg_wd = Configuration({"optimizer_attributes": {"type": "AdamW", "params": {"lr": 1e-3, "weight_decay": 0.05}}})
    assert [g_["weight_decay"] for g_ in model.get_optimizer_parameters(cfg_wd)] == [0.05, 0.05, 0.0, 0.0]
    names = {id(p): n for n, p in model.named_parameters()}
    assert all(("bias" in names[id(p)] or "LayerNorm" in names[id(p)]) for g_ in (pg[2], pg[3]) for p in g_["params"])
    assert all("img_encoder." in names[id(p)] or "text_encoder.e" in names[id(p)] for p in pg[0]["params"])
    assert any("text_projection" in names[id(p)] for p in pg[1]["params"])  # not a tower prefix: trains at the full lr
    assert sum(len(g_["params"]) for g_ in pg) == len(list(model.parameters()))
    opt = build_optimizer(model, cfg, use_hip_arena=True)
    assert len(opt.arena.groups) == len([g_ for g_ in pg if g_["params"]])
    out = model(sl)
    loss = out["losses"]["level1_similarity_loss"]
    ref = float(g["b4n1.loss"])
    assert abs(float(loss) - ref) <= 1e-3 * abs(ref), (float(loss), ref)
    before = {id(p): p.detach().clone() for p in model.parameters()}
    loss.backward()
    opt.step()
    moved = []
    for g_ in pg:
        lr = g_.get("lr", 1e-3)
        for p in g_["params"]:
            d = (p.detach() - before[id(p)]).abs()
            nz = d[d > 0]
            if nz.numel() < 8:
                continue   # a parameter without gradient on this batch (e.g. unused position rows)
            # first AdamW step, no weight decay: |delta| = lr |g| / (|g| + eps) <= lr, and = lr wherever |g| >> eps
            assert float(d.max()) <= lr * 1.001 + 1e-12, (names[id(p)], float(d.max()), lr)
            moved.append((float(nz.median()) / lr, names[id(p)]))
    assert len(moved) >= 20 and sum(1 for m_, _ in moved if m_ > 0.5) >= 0.8 * len(moved), sorted(moved)[:5]
    return dict(loss=float(loss), ref=ref, groups=[len(g_["params"]) for g_ in pg], moved=len(moved))


def case_bert_layer_dropout(dev):
    """Fused BERT layer in training mode with attention / hidden dropout vs the oracle layer with the SAME masks (rebuilt on the
    host by the numpy twin of the counter-based hash): forward, input gradient, parameter gradients."""
    import numpy as np

    from antmmf.hip import functional as HF
    from antmmf.modules.vision.backbone.clip.configuration_bert import BertConfig
    from antmmf.modules.vision.backbone.clip.modeling_bert import BertLayer
    from kernel_cases import dropout_keep_np
    from oracle import towers as otowers

    pa, ph, seed = 0.2, 0.1, (31 << 32) | 2718
    cfg = BertConfig(vocab_size_or_config_json_file=50, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=512,
                     hidden_act="gelu", hidden_dropout_prob=ph, attention_probs_dropout_prob=pa, layer_norm_eps=1e-12)
    layer = BertLayer(cfg)
    W.fill_module_(layer)
    layer = layer.to(dev).train()
    B, N, d, heads = 3, 12, 128, 2
    x0 = W.data_tensor("bertdrop.x", (B, N, d))
    w = W.data_tensor("bertdrop.w", (B, N, d))
    lengths = torch.tensor([12, 7, 3])
    key_bias = (1.0 - (torch.arange(N)[None, :] < lengths[:, None]).float()) * -10000.0
    x = x0.to(dev, torch.bfloat16).requires_grad_(True)
    y = HF.transformer_layer(x, layer._spec_train, layer._params(), key_bias.to(dev), seed=seed)
    (y.float() * w.to(dev)).sum().backward()
    drop = dict(attn=dropout_keep_np(np.arange(B * heads * N * N).reshape(B, heads, N, N), seed, pa).float() / (1 - pa),
                hid1=dropout_keep_np(np.arange(B * N * d).reshape(B, N, d), seed + 1, ph).float() / (1 - ph),
                hid2=dropout_keep_np(np.arange(B * N * d).reshape(B, N, d), seed + 2, ph).float() / (1 - ph))
    P = {n: p.detach().float().cpu().clone().requires_grad_(True) for n, p in layer.named_parameters()}
    xr = x0.to(torch.bfloat16).float().requires_grad_(True)
    ref = otowers.bert_layer(P, xr, key_bias, heads, drop=drop)
    (ref * w).sum().backward()
    check("bertdrop.y", y, ref, 5e-2, 3e-2)
    check("bertdrop.dx", x.grad, xr.grad, 1e-1, 5e-2)
    named = dict(layer.named_parameters())
    top = max(float(v.grad.norm()) for v in P.values())
    n = 0
    for k, v in P.items():
        rn = float(v.grad.norm())
        if rn > 1e-4 * top:
            assert abs(float(named[k].grad.float().norm()) - rn) <= 0.1 * rn, (k, float(named[k].grad.float().norm()), rn)
            n += 1
    assert n >= 14
    # eval mode: no dropout, deterministic
    layer.eval()
    y1 = layer(x.detach(), key_bias.to(dev))
    y2 = layer(x.detach(), key_bias.to(dev))
    assert torch.equal(y1, y2)
    return dict(checked=n)


def case_temporal_head(dev, golden=None, hidden=128, heads=2, bsz=3, n_clips=8):
    """(hidden / heads / bsz / n_clips other than the defaults: the same check at config 3's REAL width -- d = 768, 12 heads, 8 clips -- against the oracle only;
    the reference fixture is at the default dims.)
    UnivlForVideo.get_temporal_output ([cls] + clip features through a 3-layer BERT, inputs_embeds path) vs the CPU oracle's
    BERT restatement on the same weights, forward and input / parameter gradients -- and, with `golden`, vs the run of the reference's own
    BERT modules over the same method body (tests/golden/ops_temporal_head.pt: output, input gradients, every parameter's full gradient)."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from oracle import towers as otowers
    from roi_univl.univl.model.univl_video_pretrain import UnivlForVideo

    tenc = dict(type="RobertBertEncoder", params=dict(pretrained=False, vocab_size=40, hidden_size=hidden, intermediate_size=4 * hidden,
                                                      num_hidden_layers=3, num_attention_heads=heads, max_position_embeddings=40,
                                                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, out_dim=hidden, is_proj=False))
    model = UnivlForVideo(Configuration(dict(TINY_CLIP_CFG, hidden_size=hidden, with_temporal_encoder=True, temporal_encoder=tenc)))
    W.fill_module_(model)
    model = model.to(dev).train()
    tag = "temporal" if (hidden, bsz, n_clips) == (128, 3, 8) else f"temporal.{hidden}.{bsz}.{n_clips}"
    clip = (W.data_tensor(tag + ".clip", (bsz, n_clips, hidden)) * 0.5)
    w = W.data_tensor(tag + ".w", (bsz, n_clips + 1, hidden))
    x = clip.detach().clone().to(dev).requires_grad_(True)
    out = model.get_temporal_output(x)
    (out.float() * w.to(dev)).sum().backward()
    P = {n: p.detach().float().cpu().clone().requires_grad_(True) for n, p in model.named_parameters() if n.startswith(("temporal_encoder.", "cls_token"))}
    xr = clip.detach().clone().requires_grad_(True)
    emb_in = torch.cat([P["cls_token"].expand(bsz, -1, -1), xr], 1)
    Pe = {k[len("temporal_encoder.embeddings."):]: v for k, v in P.items() if k.startswith("temporal_encoder.embeddings.")}
    Pe["word_embeddings.weight"] = None
    h = otowers.bert_embeddings(Pe, inputs_embeds=emb_in)
    ref = otowers.bert_encoder({k[len("temporal_encoder.encoder."):]: v for k, v in P.items() if k.startswith("temporal_encoder.encoder.")},
                               h, torch.zeros(bsz, n_clips + 1), heads)
    (ref * w).sum().backward()
    check("temporal.out", out, ref, 5e-2, 3e-2)
    check("temporal.dclip", x.grad, xr.grad, 1e-1, 5e-2)
    check("temporal.dcls", model.cls_token.grad, P["cls_token"].grad, 1e-1, 5e-2)
    named = dict(model.named_parameters())
    n = 0
    top = max(float(v.grad.norm()) for v in P.values() if v.grad is not None)
    for k, v in P.items():
        if v.grad is None or named[k].grad is None:
            continue
        rn = float(v.grad.norm())
        if rn > 1e-4 * top:  # (key-projection biases have an analytically zero gradient)
            assert abs(float(named[k].grad.float().norm()) - rn) <= 0.15 * rn, (k, float(named[k].grad.float().norm()), rn)
            n += 1
    assert n > 30
    res = dict(checked=n)
    if golden is None:   # oracle only: directions as well (cosine per parameter against the oracle's gradients)
        rows = []
        for k, v in P.items():
            if v.grad is not None and named[k].grad is not None and float(v.grad.norm()) > 1e-4 * top:
                a, b = named[k].grad.detach().float().flatten().cpu(), v.grad.flatten()
                rows.append((float(torch.dot(a, b) / (a.norm() * b.norm())), k))
        rows.sort()
        assert rows[0][0] >= 0.995, rows[:4]
        res["min_cos"] = rows[0]
    if golden is not None:
        g = golden("ops_temporal_head.pt")
        check("temporal.out.ref", out, g["out"], 5e-2, 3e-2)
        check("temporal.dclip.ref", x.grad, g["dclip"], 1e-1, 5e-2)
        check("temporal.dcls.ref", model.cls_token.grad, g["dcls"], 1e-1, 5e-2)
        torch.testing.assert_close(ref.detach(), g["out"], rtol=2e-5, atol=2e-5)   # the oracle itself is pinned by the same fixture
        temporal = [(k, v) for k, v in model.named_parameters() if k.startswith(("temporal_encoder.", "cls_token"))]
        res["directions"] = assert_grad_directions(temporal, g, "", min_checked=30)
    return res


def case_univl_stage2(dev, golden, mining=False):
    """stage1 + stage2 of the product model.  Plain: against the reference run (e2e_clip_stage2.pt).  Hard-negative mining +
    mean re-weighting: against the CPU oracle given the video indices the model actually picked (torch.topk(sorted=False) order
    is unspecified, so the reference's own pick order cannot be pinned)."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    g = golden("e2e_clip_stage2.pt")
    extra = dict(hard_example_mining=True, re_sample_method="top_k", re_weight_method="median") if mining else {}
    cfg = Configuration(dict(TINY_CLIP_CFG, training_stage="stage1+stage2", with_cross_encoder=True, **extra))
    model = UnivlForVideoTextRetrieval(cfg)
    W.fill_module_(model)
    model = model.to(dev).train()
    model.dropout.p = 0.0
    img, ids, mask = g["s2.image_data"].to(dev), g["s2.input_ids"].to(dev), g["s2.input_mask"].to(dev)
    bsz, n_clips = img.shape[0], 2
    img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                     image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
    cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
    out = model(img_input, cap_input)
    l1, l2 = out["losses"]["level1_similarity_loss"], out["losses"]["level2_similarity_loss"]
    if not mining:
        out["l2_simi"].retain_grad()   # (reduce_clips is the identity at level 2: this is the matrix the level-2 loss reads)
    (l1 + l2).backward(retain_graph=not mining)
    if not mining:
        loss_grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        c_prod = out["l2_simi"].grad.detach().float().cpu().clone()
        # (1) THE gradient check of this path: a scalar of the same graph whose gradient does not cancel -- fixed POSITIVE weights on the
        # cross-encoder pair scores (tests/golden/make_golden.py gen_e2e_clip_stage2, "s2.pin"): every parameter of both towers, the cross
        # encoder and the score head against the reference's gradient at the gates of stage 1 / stage 3 / M2 (measured on MI355X: min cosine
        # 0.9978, worst norm 1.7 % over 72 parameters)
        model.zero_grad(set_to_none=True)
        pin = (out["l2_simi"].float() * (W.data_tensor("s2.pin", tuple(out["l2_simi"].shape)).abs() + 0.5).to(dev)).sum()
        pin.backward(retain_graph=True)
        assert abs(float(pin) - float(g["s2.pin.value"])) <= 5e-2 * max(1.0, abs(float(g["s2.pin.value"]))), (float(pin), float(g["s2.pin.value"]))
        pin_dirs = assert_grad_directions(model.named_parameters(), g, "s2.pin.", min_cos=0.995, max_norm_rel=0.05, min_checked=50)
        prow = [(p.grad.detach().float().flatten().cpu(), g[f"s2.pin.gfull.{n}"].float().flatten()) for n, p in model.named_parameters()
                if p.grad is not None and f"s2.pin.gfull.{n}" in g]
        pgot, pref = torch.cat([r[0] for r in prow]), torch.cat([r[1] for r in prow])
        pin_dirs["global_cos"] = float(torch.dot(pgot, pref) / (pgot.norm() * pref.norm()))
        assert pin_dirs["global_cos"] >= 0.999, pin_dirs
        # (1b) the level-2 LOSS gradient, checked tightly in two exact steps.  loss gradient = J^T c with c = d loss / d l2_simi:
        #   * c itself: the loss kernel's gradient at the product's own scores against the oracle's formula there (fp32: 1e-6), and against the
        #     reference's c (the scores carry bf16 noise: 3 %);
        #   * J^T (the backward of both towers, the cross encoder and the score head -- linear in its upstream gradient) applied to c+ = max(c, 0) and
        #     c- = max(-c, 0) of the REFERENCE's c: two scalars with non-negative weights, no cancellation, each at the pin's gates.  J^T c = J^T c+ - J^T c-
        #     exactly; the 13x cancellation between the halves (norms 18.7 / 18.7 -> 1.4) is what makes the direct comparison (2) a noise measurement.
        from oracle import losses as olosses_s2

        s_here = out["l2_simi"].detach().float().cpu().clone().requires_grad_(True)
        olosses_s2.mil_nce(s_here, bsz, 1, None).backward()
        assert float((c_prod - s_here.grad).abs().max()) <= 1e-6, (c_prod, s_here.grad)
        c_ref = g["s2.plain.dl2_simi"].float()
        assert float((c_prod - c_ref).abs().max()) <= 3e-2 * float(c_ref.abs().max()), (c_prod, c_ref)
        half_dirs = {}
        for sign, key in ((1.0, "pinp"), (-1.0, "pinm")):
            model.zero_grad(set_to_none=True)
            (out["l2_simi"].float() * (sign * c_ref).clamp(min=0).to(dev)).sum().backward(retain_graph=True)
            # (gates: the pin's, a notch wider on the direction -- MI355X measured min cosine 0.987 on the patch-embedding weight for c+, whose weights sit on
            # the off-diagonal pairs only, worst norm 5.1 % on a key projection; lane emulator 0.9985 / 0.9999, 1.3 %)
            dkey = assert_grad_directions(model.named_parameters(), g, f"s2.{key}.", min_cos=0.98, max_norm_rel=0.08, min_checked=50)
            hrow = [(p.grad.detach().float().flatten().cpu(), g[f"s2.{key}.gfull.{n}"].float().flatten()) for n, p in model.named_parameters()
                    if p.grad is not None and f"s2.{key}.gfull.{n}" in g]
            hgot, href = torch.cat([r[0] for r in hrow]), torch.cat([r[1] for r in hrow])
            dkey["global_cos"] = float(torch.dot(hgot, href) / (hgot.norm() * href.norm()))
            assert dkey["global_cos"] >= 0.998, (key, dkey)
            half_dirs[key] = dkey
        for n, p in model.named_parameters():   # the checks below are on the LOSS gradient again
            p.grad = loss_grads.get(n)
        ref1, ref2 = float(g["s2.plain.loss1"]), float(g["s2.plain.loss2"])
        assert abs(float(l1) - ref1) <= 1e-3 * abs(ref1), (float(l1), ref1)
        assert abs(float(l2) - ref2) <= 2e-3 * abs(ref2), (float(l2), ref2)   # cross-encoder scores through 2 more bf16 layers + an MLP
        check("s2.l2_simi", out["l2_simi"], g["s2.plain.l2_simi"], 5e-2, 3e-2)
        worst = []
        for n, p in model.named_parameters():
            key = f"s2.plain.gnorm.{n}"
            if key in g and p.grad is not None:
                worst.append((abs(float(p.grad.float().norm()) - float(g[key])), float(g[key]), n))
        top = max(w[1] for w in worst)
        rel = sorted(((w[0] / w[1], w[2]) for w in worst if w[1] > 1e-4 * top), reverse=True)
        # (2) the LOSS gradient: a noise measurement more than a parity check.  The level-2 loss is a softmax over pair scores that are nearly the
        # same function of the shared tower weights at random init; its gradient rows sum to zero, so only the per-pair DEVIATION of
        # d score / d theta counts (~1 % of the common part) and bf16's 0.2-0.4 % rounding of the common part is tens of per cent of it.  Two
        # correct builds that differ only in the fp32 summation order inside LayerNorm measured whole-model cosine 0.96 and 0.80 on MI355X (per
        # parameter 0.84 / 0.64, norms off by 15 % / 41 %), the lane emulator 0.96 -- while (1), the same backward code on the same graph,
        # sits at 0.998.  The gates below only catch gross errors.
        assert len(rel) > 50 and rel[0][0] < 0.6, rel[:5]
        dirs = assert_grad_directions(model.named_parameters(), g, "s2.plain.", min_cos=0.4, max_norm_rel=0.6, min_checked=50)
        rows = [(p.grad.detach().float().flatten().cpu(), g[f"s2.plain.gfull.{n}"].float().flatten()) for n, p in model.named_parameters()
                if p.grad is not None and f"s2.plain.gfull.{n}" in g]
        got, ref = torch.cat([r[0] for r in rows]), torch.cat([r[1] for r in rows])
        dirs["global_cos"] = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
        assert dirs["global_cos"] >= 0.7, dirs
        return dict(loss1=(float(l1), ref1), loss2=(float(l2), ref2), worst=rel[:3], directions=dirs, pin_directions=pin_dirs, half_directions=half_dirs)
    from oracle import step as ostep

    P = tiny_models.clip_arch_params(stage2=True)
    chosen = model._last_chosen.cpu()
    l1m = out["l1_simi"].detach().float().cpu()
    d = torch.diagonal(l1m)
    weight = torch.where(d > d.mean(), torch.clamp((d.mean() - d.min()) / (d - d.min()), min=0.2), torch.ones_like(d))
    ref = ostep.univl_stage2(P, g["s2.image_data"], g["s2.input_ids"], g["s2.input_mask"], n_clips, vit_heads=2, patch=8, bert_heads=2,
                             chosen=chosen, weight=weight)
    assert bool((torch.diagonal(chosen) == torch.arange(bsz)).all()) and chosen.shape == (bsz, bsz)
    check("s2.mine.l2_simi", out["l2_simi"], ref["l2_simi"], 5e-2, 3e-2)
    assert abs(float(l2) - float(ref["loss"])) <= 5e-3 * abs(float(ref["loss"])), (float(l2), float(ref["loss"]))
    return dict(loss2=(float(l2), float(ref["loss"])))


def case_univl_stage2_cnvid_gate(dev, golden):
    """The whole product model under the CN-VID schedule (config carries change_iter / change_rate; the batch carries `incre_num` as the trainer writes it):
    incre_num 0 -> the plain cross-encoder branch WITH the row re-weighting, level-2 loss against the reference's cnvid_vtp class on the same batch
    (tests/golden/e2e_cnvid_gate.pt "plain.loss2"); incre_num 1 -> the mined branch (every draw / 100 < 1)."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from antmmf.structures.sample import SampleList
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    g, gate = golden("e2e_clip_stage2.pt"), golden("e2e_cnvid_gate.pt")
    cfg = Configuration(dict(TINY_CLIP_CFG, training_stage="stage1+stage2", with_cross_encoder=True, hard_example_mining=True, re_sample_method="top_k",
                             re_weight_method="median", change_iter=5000, change_rate=0.15))
    model = UnivlForVideoTextRetrieval(cfg)
    W.fill_module_(model)
    model = model.to(dev).train()
    model.dropout.p = 0.0
    img, ids, mask = g["s2.image_data"].to(dev), g["s2.input_ids"].to(dev), g["s2.input_mask"].to(dev)
    bsz, n_clips = img.shape[0], 2
    img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                     image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
    cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
    with torch.no_grad():
        out = model(img_input, cap_input, sample_list=SampleList(incre_num=0.0))
        assert model._last_mined is False
        l2, ref = float(out["losses"]["level2_similarity_loss"]), float(gate["plain.loss2"])
        assert abs(l2 - ref) <= 2e-3 * abs(ref), (l2, ref)   # the level-2 gate of case_univl_stage2
        model(img_input, cap_input, sample_list=SampleList(incre_num=1.0))
        assert model._last_mined is True
        model(img_input, cap_input)   # no incre_num in the batch: the reference signature's default 0.0 -> never mined
        assert model._last_mined is False
    return dict(loss2=(l2, ref))


def moco_queue(name, dim, K):
    return torch.nn.functional.normalize(W.data_tensor(name, (dim, K)), dim=0)


def case_univl_moco(dev, golden, with_optimizer=False):
    """Product UnivlForVideoTextRetrieval with MoCo (momentum key towers, queues, fused MoCo loss) over the two steps the
    reference was run for (tests/golden/make_golden.py gen_e2e_clip_moco; K=64, M=0.5; weights x1.05 between the steps).
    with_optimizer=True puts the parameters into the flat HipAdamW arena first, so the key towers mirror the arena and the
    momentum update is the single fused launch."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.moco_utils import MocoUtils
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    g = golden("e2e_clip_moco.pt")
    tag, n_clips = "moco", 2
    cfg = Configuration(dict(TINY_CLIP_CFG, with_moco=True, K=64, M=0.5))
    model = UnivlForVideoTextRetrieval(cfg)
    W.fill_module_(model)
    model = model.to(dev).train()
    opt = None
    if with_optimizer:
        from antmmf.hip.arena import HipAdamW

        opt = HipAdamW([{"params": [p for p in model.parameters() if p.requires_grad]}], lr=0.0)
    mu = MocoUtils(cfg, img_encoder=model.module.img_encoder, txt_encoder=model.module.text_encoder).to(dev)
    mu.txt_queue.copy_(moco_queue("moco.txt_queue", 128, 64))
    mu.img_queue.copy_(moco_queue("moco.img_queue", 128, 16384))
    model.moco_utils = mu
    img = g[f"{tag}.image_data"].to(dev)
    ids, mask = g[f"{tag}.input_ids"].to(dev), g[f"{tag}.input_mask"].to(dev)
    bsz = img.shape[0]
    img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                     image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
    cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
    res = {}
    for stp in (1, 2):
        if opt is not None:
            opt.zero_grad()
        else:
            model.zero_grad(set_to_none=True)
        out = model(img_input, cap_input)
        loss = out["losses"]["level1_similarity_loss"]
        ref_loss = float(g[f"{tag}.loss{stp}"])
        # temperature 0.05 multiplies the bf16 towers' similarity error by 20: 5e-3 relative on a loss of ~9.8
        assert abs(float(loss) - ref_loss) <= 5e-3 * abs(ref_loss), (stp, float(loss), ref_loss)
        loss.backward()
        worst = []
        for n, p in model.named_parameters():
            key = f"{tag}.gnorm{stp}.{n}"
            if key not in g or p.grad is None:
                continue
            worst.append((abs(float(p.grad.float().norm()) - float(g[key])), float(g[key]), n))
        top = max(w[1] for w in worst)
        rel = sorted(((w[0] / w[1], w[2]) for w in worst if w[1] > 1e-4 * top), reverse=True)
        assert len(rel) > 40 and rel[0][0] < 0.2, rel[:5]
        check(f"{tag}.txt_queue{stp}", mu.txt_queue[:, :12], g[f"{tag}.txt_queue_head{stp}"], 5e-2, 3e-2)
        check(f"{tag}.img_queue{stp}", mu.img_queue[:, :20], g[f"{tag}.img_queue_head{stp}"], 5e-2, 3e-2)
        assert int(mu.txt_queue_ptr) == int(g[f"{tag}.txt_ptr{stp}"]) and int(mu.img_queue_ptr) == int(g[f"{tag}.img_ptr{stp}"])
        res[f"loss{stp}"] = (float(loss), ref_loss)
        if stp == 1:
            with torch.no_grad():
                if opt is not None:
                    opt.arena.master.mul_(1.05)
                    opt.arena.sync_shadow()
                else:
                    for p in model.module.parameters():
                        p.mul_(1.05)
    kq = dict(mu.txt_encoder_k.named_parameters())["encoder.layer.0.attention.self.query.weight"]
    check(f"{tag}.key_probe", kq[:4, :8], g[f"{tag}.key_probe"], 1e-4, 1e-5)
    if opt is not None:
        assert mu._flat is not None, "key towers should mirror the optimizer arena"
    return res


def case_univl_stage2_loss_contract(dev, k=6):
    """The 1e-3 loss contract on the stage-2 (cross-encoder) step over k seeded batches against the oracle (pinned to the reference's stage-2 run, e2e_clip_stage2.pt): mean
    relative deviation of the level-1 and of the level-2 loss <= 1e-3; every single batch at the single-batch gates of case_univl_stage2 (1e-3 / 2e-3)."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from oracle import step as ostep
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    model = UnivlForVideoTextRetrieval(Configuration(dict(TINY_CLIP_CFG, training_stage="stage1+stage2", with_cross_encoder=True)))
    W.fill_module_(model)
    model = model.to(dev).train()
    model.dropout.p = 0.0
    P = tiny_models.clip_arch_params(stage2=True)
    bsz, n_clips, rel1, rel2 = 4, 2, [], []
    for b in range(k):
        img = W.data_tensor(f"s2k.image.{b}", (bsz, n_clips, 3, 32, 32))
        lengths = W.data_ints(f"s2k.len.{b}", (bsz,), 3, 13)
        mask = (torch.arange(12)[None, :] < lengths[:, None]).long()
        ids = W.data_ints(f"s2k.ids.{b}", (bsz, 12), 1, 300) * mask
        ids[:, 0] = 101
        with torch.no_grad():
            out = model(dict(image_data=img.to(dev), image_pad_mask=torch.zeros(bsz, n_clips, 32, 32, dtype=torch.bool, device=dev), image_n_clips=[n_clips] * bsz,
                             image_num_frames=[1] * bsz), dict(caption_input_ids=ids.to(dev), caption_input_mask=mask.to(dev), caption_raw_input_ids=ids.to(dev)))
            r1 = float(ostep.univl_stage1(P, img, ids, mask, n_clips, 2, 8, 2)["loss"])
            r2 = float(ostep.univl_stage2(P, img, ids, mask, n_clips, 2, 8, 2)["loss"])
        rel1.append((float(out["losses"]["level1_similarity_loss"]) - r1) / abs(r1))
        rel2.append((float(out["losses"]["level2_similarity_loss"]) - r2) / abs(r2))
    m1, m2 = sum(rel1) / k, sum(rel2) / k
    if os.environ.get("ANTMMF_REAL_WIDTH_OUT"):
        import json

        with open(os.environ["ANTMMF_REAL_WIDTH_OUT"], "a") as f:
            f.write(json.dumps(dict(case="stage2_loss_contract", level1_rel=rel1, level2_rel=rel2, mean=[m1, m2])) + "\n")
    assert abs(m1) <= 1e-3 and max(abs(r) for r in rel1) <= 1e-3, (m1, rel1)
    assert abs(m2) <= 1e-3 and max(abs(r) for r in rel2) <= 2e-3, (m2, rel2)
    return dict(mean=(m1, m2), level2_rel=rel2)


def case_univl_moco_loss_contract(dev, k=6):
    """north_star's "loss within 1e-3 rel" tested on what it states instead of on one batch's noise floor: the MoCo step's loss on k seeded
    batches against the oracle (pinned to the reference's MoCo run at 1e-5, tests/test_oracle_golden.py) -- the MEAN relative deviation must meet the contract,
    every single batch the single-batch gate of case_univl_moco (temperature 0.05 multiplies the bf16 towers' similarity error by 20)."""
    import roi_univl  # noqa: F401
    from antmmf.common.configuration import Configuration
    from oracle import step as ostep
    from roi_univl.univl.model.moco_utils import MocoUtils
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    n_clips, bsz, rel = 2, 4, []
    cfg = Configuration(dict(TINY_CLIP_CFG, with_moco=True, K=64, M=0.5))
    for b in range(k):
        model = UnivlForVideoTextRetrieval(cfg)
        W.fill_module_(model)
        model = model.to(dev).train()
        mu = MocoUtils(cfg, img_encoder=model.module.img_encoder, txt_encoder=model.module.text_encoder).to(dev)
        mu.txt_queue.copy_(moco_queue("moco.txt_queue", 128, 64))
        mu.img_queue.copy_(moco_queue("moco.img_queue", 128, 16384))
        model.moco_utils = mu
        img = W.data_tensor(f"mocok.image.{b}", (bsz, n_clips, 3, 32, 32))
        lengths = W.data_ints(f"mocok.len.{b}", (bsz,), 3, 13)
        mask = (torch.arange(12)[None, :] < lengths[:, None]).long()
        ids = W.data_ints(f"mocok.ids.{b}", (bsz, 12), 1, 300) * mask
        ids[:, 0] = 101
        with torch.no_grad():
            out = model(dict(image_data=img.to(dev), image_pad_mask=torch.zeros(bsz, n_clips, 32, 32, dtype=torch.bool, device=dev), image_n_clips=[n_clips] * bsz,
                             image_num_frames=[1] * bsz), dict(caption_input_ids=ids.to(dev), caption_input_mask=mask.to(dev), caption_raw_input_ids=ids.to(dev)))
            P = tiny_models.clip_arch_params()
            queues = dict(txt=moco_queue("moco.txt_queue", 128, 64), img=moco_queue("moco.img_queue", 128, 16384), txt_ptr=0, img_ptr=0)
            ref = float(ostep.univl_stage1_moco(P, {kk: v.clone() for kk, v in P.items()}, queues, img, ids, mask, n_clips, vit_heads=2, patch=8, bert_heads=2,
                                                momentum=0.5, temperature=0.05)["loss"])
        rel.append((float(out["losses"]["level1_similarity_loss"]) - ref) / abs(ref))
    mean = sum(rel) / len(rel)
    if os.environ.get("ANTMMF_REAL_WIDTH_OUT"):   # tools/gpu_*.sh: the measured deviations next to the other reports of the run
        import json

        with open(os.environ["ANTMMF_REAL_WIDTH_OUT"], "a") as f:
            f.write(json.dumps(dict(case="moco_loss_contract", rel=rel, mean_rel=mean)) + "\n")
    assert abs(mean) <= 1e-3 and max(abs(r) for r in rel) <= 5e-3, (mean, rel)
    return dict(mean_rel=mean, rel=rel)


def load_dmae_utils():
    """prj/dmae_vtp's dmae_utils module, by path (its package is also called roi_univl, like base_vtp's)."""
    import importlib.util

    path = os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "dmae_vtp", "roi_univl", "univl", "model", "dmae_utils.py")
    spec = importlib.util.spec_from_file_location("antmmf_dmae_utils", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DMAE_CFG = dict(hidden_size=128, l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="seqTransf", l3_partial_type=-1,
                l3_max_frames=6, l3_max_words=12, l3_sim_header_hidden_layer=2)


def case_dmae_seqtransf(dev, golden):
    """DmaeUtils._agg_visual_feat(seqTransf) -- frame position embedding + 2 masked CLIP blocks (fused HIP layers) + residual --
    vs the reference run (ops_dmae_seqtransf.pt), forward and every parameter gradient."""
    from antmmf.common.configuration import Configuration

    g = golden("ops_dmae_seqtransf.pt")
    du = load_dmae_utils().DmaeUtils(Configuration(DMAE_CFG))
    W.fill_module_(du)
    du = du.to(dev)
    x = g["visual"].detach().clone().to(dev).requires_grad_(True)   # (never flag the cached fixture tensor itself)
    out, tok_mask, orig = du._agg_visual_feat(x, g["mask"].to(dev), "seqTransf")
    check("dmae.out", out, g["out"], 5e-2, 3e-2)
    check("dmae.tok_mask", tok_mask, g["tok_mask"], 0, 0)
    (out.float() * g["w"].to(dev)).sum().backward()
    check("dmae.dvisual", x.grad, g["dvisual"], 1e-1, 5e-2)
    n = 0
    for name, p in du.named_parameters():
        if "grad." + name in g:
            check("dmae.grad." + name, p.grad, g["grad." + name], 1e-1, 1e-1)
            n += 1
        elif "gnorm." + name in g:
            ref = float(g["gnorm." + name])
            assert abs(float(p.grad.float().norm()) - ref) <= 0.15 * ref, (name, float(p.grad.float().norm()), ref)
            n += 1
    assert n >= 25
    return dict(checked=n)


DMAE_E2E = dict(l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="meanP", l3_partial_type=-1, l3_max_frames=4,
                l3_max_words=12, l3_sim_header_hidden_layer=2)


def case_dmae_stage3_tpm(dev_str):
    """Same model with TPM-CL on (l3_partial_type 4): level-3 loss and the gradient norms of the new head / the tower projections."""
    return r"""
import os, sys, torch
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "ant-multi-modal-framework_amd"),
                os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "dmae_vtp"), ROOT]
import weightgen as W
import roi_univl
from antmmf.common.configuration import Configuration
from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval
dev = torch.device(%r)
g = torch.load(os.path.join(ROOT, "tests", "golden", "e2e_dmae_stage3.pt"))
model = UnivlForVideoTextRetrieval(Configuration(dict(%r, training_stage="stage1+stage3", l3_loss_type="negNCE", **dict(%r, l3_partial_type=4))))
W.fill_module_(model)
model.dmae_utils.tis_selector.thresh.fill_(0.6)
model = model.to(dev).train()
img, ids, mask = g["s3.image_data"].to(dev), g["s3.input_ids"].to(dev), g["s3.input_mask"].to(dev)
bsz, n_clips = img.shape[0], 4
img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                 image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
out = model(img_input, cap_input)
l3 = out["losses"]["level3_similarity_loss"]
(out["losses"]["level1_similarity_loss"] + l3).backward()
r3 = float(g["s3.tpm4.loss3"])
assert abs(float(l3) - r3) <= 2e-2 * abs(r3), (float(l3), r3)
named = dict(model.named_parameters())
bad = []
for k in g:
    if k.startswith("s3.tpm4.gnorm."):
        n = k[len("s3.tpm4.gnorm."):]
        ref = float(g[k]); got = float(named[n].grad.float().norm()) if named[n].grad is not None else 0.0
        if ref > 1e-4 and abs(got - ref) > 0.25 * ref:
            bad.append((n, got, ref))
assert not bad, bad
print("okdmae", float(l3), r3)
""" % (ROOT, dev_str, TINY_CLIP_CFG, DMAE_E2E)


def case_dmae_stage3_loss_contract(dev_str, k=6):
    """The loss contract of north_star on k seeded batches of the DMAE step (stage 1 + stage 3) against the oracle (pinned to the reference's stage-3 run, e2e_dmae_stage3.pt, in
    tests/test_oracle_golden.py).  Level 1: 1e-3 on every batch and on the mean.  Level 3 is a NegNCE over token-wise cosines at logit scale 100: an absolute error of 1e-3 on a
    score is 0.1 on a logit, and the ORACLE ITSELF under torch's bf16 autocast moves by 0.15 ... 1.7 % per batch with either sign on these very batches (printed next to the
    product's deviations) -- no bf16 build, the reference under its own autocast included, holds 1e-3 there, and this build, whose token features live in bf16 between the
    towers and the head, is 0.4 - 0.8 % off (same sign on most batches).  STATED, not hidden: level 3 of DMAE does not meet the 1e-3 contract; it is held to what is measured
    -- the scores to 3e-3 absolute (magnitude 0.1 - 0.17), the loss RMS over the batches to max(8e-3, 1.5 x the autocast oracle's own RMS on the same batches, measured in
    the same run: 0.78 % on these six) and 2.5e-2 on any batch (the autocast oracle's worst batch: 1.7e-2).
    Subprocess code, like case_dmae_stage3 (dmae_vtp's package is also called roi_univl)."""
    return r"""
import os, sys, torch
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "ant-multi-modal-framework_amd"),
                os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "dmae_vtp"), ROOT]
import weightgen as W, tiny_models
import roi_univl
from antmmf.common.configuration import Configuration
from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval
from oracle import step as ostep
dev = torch.device(%r)
K = %d
model = UnivlForVideoTextRetrieval(Configuration(dict(%r, training_stage="stage1+stage3", l3_loss_type="negNCE", **%r)))
W.fill_module_(model)
model = model.to(dev).train()
P = tiny_models.clip_arch_params(dmae=True)
bsz, n_clips, rel1, rel3, floor3, serr, head3, tower3 = 4, 4, [], [], [], [], [], []
for b in range(K):
    img = W.data_tensor(f"dmaek.image.{b}", (bsz, n_clips, 3, 32, 32))
    lengths = W.data_ints(f"dmaek.len.{b}", (bsz,), 3, 13)
    mask = (torch.arange(12)[None, :] < lengths[:, None]).long()
    ids = W.data_ints(f"dmaek.ids.{b}", (bsz, 12), 1, 300) * mask
    ids[:, 0] = 101
    with torch.no_grad():
        out = model(dict(image_data=img.to(dev), image_pad_mask=torch.zeros(bsz, n_clips, 32, 32, dtype=torch.bool, device=dev), image_n_clips=[n_clips] * bsz,
                         image_num_frames=[1] * bsz), dict(caption_input_ids=ids.to(dev), caption_input_mask=mask.to(dev), caption_raw_input_ids=ids.to(dev)))
        r1 = float(ostep.univl_stage1(P, img, ids, mask, n_clips, 2, 8, 2)["loss"])
        o3 = ostep.dmae_stage3(P, img, ids, mask, n_clips, 2, 8, 2, loss_type="negNCE", sim_header="meanP")
        with torch.autocast("cpu", dtype=torch.bfloat16):
            a3 = float(ostep.dmae_stage3(P, img, ids, mask, n_clips, 2, 8, 2, loss_type="negNCE", sim_header="meanP")["loss"])
        # which part of the level-3 deviation is the head's and which the towers': the ORACLE's fp32 head evaluated on the PRODUCT's tower outputs (word / frame token features
        # and the sentence embedding, as handed to DmaeUtils.get_similarity_logits)
        img_in = dict(image_data=img.to(dev), image_pad_mask=torch.zeros(bsz, n_clips, 32, 32, dtype=torch.bool, device=dev), image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
        cap_in, vis_in, _, _ = model.module.get_l2_input(img_in, dict(caption_input_ids=ids.to(dev), caption_input_mask=mask.to(dev), caption_raw_input_ids=ids.to(dev)))
        h3 = float(ostep.dmae_stage3_head(P, cap_in[0].float().cpu(), vis_in[0].float().cpu(), vis_in[1].float().cpu(), cap_in[2].float().cpu(), mask,
                                          loss_type="negNCE", sim_header="meanP")["loss"])
    r3 = float(o3["loss"])
    rel1.append((float(out["losses"]["level1_similarity_loss"]) - r1) / abs(r1))
    rel3.append((float(out["losses"]["level3_similarity_loss"]) - r3) / abs(r3))
    floor3.append((a3 - r3) / abs(r3))
    head3.append((float(out["losses"]["level3_similarity_loss"]) - h3) / abs(r3))   # product head vs fp32 head, same (product) features
    tower3.append((h3 - r3) / abs(r3))                                             # fp32 head: product features vs fp32 features
    serr.append(float((out["l3_simi"].float().cpu() - o3["l3_simi"]).abs().max()))
m1 = sum(rel1) / K
rms = lambda v: (sum(x * x for x in v) / len(v)) ** 0.5
rep = os.environ.get("ANTMMF_REAL_WIDTH_OUT")
if rep:
    import json
    open(rep, "a").write(json.dumps(dict(case="dmae_stage3_loss_contract", level1_rel=rel1, level3_rel=rel3, level3_rel_autocast_oracle=floor3, rms=[rms(rel3), rms(floor3)], scores_max_abs=max(serr),
                                         level3_head_part=head3, level3_tower_part=tower3, rms_parts=[rms(head3), rms(tower3)])) + "\n")
assert abs(m1) <= 1e-3 and max(abs(r) for r in rel1) <= 1e-3, (m1, rel1)
assert max(serr) <= 3e-3, serr
assert rms(rel3) <= max(8e-3, 1.5 * rms(floor3)) and max(abs(r) for r in rel3) <= 2.5e-2, (rel3, floor3)
print("okdmaek", "level1 mean", m1, "level3 rel", [round(r, 5) for r in rel3], "autocast-oracle rel", [round(r, 5) for r in floor3], "rms", rms(rel3), rms(floor3), "scores max abs", max(serr),
      "head part", [round(r, 5) for r in head3], "tower part", [round(r, 5) for r in tower3], "rms parts", rms(head3), rms(tower3))
""" % (ROOT, dev_str, k, TINY_CLIP_CFG, DMAE_E2E)


def case_dmae_stage3(loss_type="negNCE"):
    """dmae_vtp product model (stage1 + stage3) vs the reference run -- executed in a subprocess because dmae_vtp's package is
    also called roi_univl (it overlays base_vtp's).  Returns the child's output; the caller asserts on "okdmae"."""
    code = r"""
import os, sys, torch
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "ant-multi-modal-framework_amd"),
                os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "dmae_vtp"), ROOT]
import weightgen as W
import roi_univl
from antmmf.common.configuration import Configuration
from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval
from kernel_cases import check
DMAE_MIN_COS = 0.995
dev = torch.device(%r)
loss_type = %r
g = torch.load(os.path.join(ROOT, "tests", "golden", "e2e_dmae_stage3.pt"))
TINY = %r
model = UnivlForVideoTextRetrieval(Configuration(dict(TINY, training_stage="stage1+stage3", l3_loss_type=loss_type, **%r)))
W.fill_module_(model)
model = model.to(dev).train()
img, ids, mask = g["s3.image_data"].to(dev), g["s3.input_ids"].to(dev), g["s3.input_mask"].to(dev)
bsz, n_clips = img.shape[0], 4
img_input = dict(image_data=img, image_pad_mask=torch.zeros(bsz, img.shape[1], 32, 32, dtype=torch.bool, device=dev),
                 image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz)
cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
out = model(img_input, cap_input)
l1, l3 = out["losses"]["level1_similarity_loss"], out["losses"]["level3_similarity_loss"]
(l1 + l3).backward()
r1, r3 = float(g[f"s3.{loss_type}.loss1"]), float(g[f"s3.{loss_type}.loss3"])
assert abs(float(l1) - r1) <= 1e-3 * abs(r1), (float(l1), r1)
# logit scale 100 on cosines computed from bf16 token features (measured 4e-3): 8e-3 on the loss
assert abs(float(l3) - r3) <= 8e-3 * abs(r3), (float(l3), r3)
check("s3.l3_simi", out["l3_simi"], g[f"s3.{loss_type}.l3_simi"], 5e-2, 5e-2)
worst = []
for n, p in model.named_parameters():
    key = f"s3.{loss_type}.gnorm.{n}"
    if key in g and p.grad is not None:
        worst.append((abs(float(p.grad.float().norm()) - float(g[key])), float(g[key]), n))
top = max(w[1] for w in worst)
rel = sorted(((w[0] / w[1], w[2]) for w in worst if w[1] > 1e-3 * top), reverse=True)
assert len(rel) > 40 and rel[0][0] < 0.05, rel[:5]
dirs = None
if loss_type == "negNCE":
    from model_cases import grad_direction_report
    rows = grad_direction_report(model.named_parameters(), g, "s3.negNCE.")
    dirs = (rows[0], max(rows, key=lambda r: r[1]), len(rows))
    assert len(rows) > 40 and rows[0][0] >= DMAE_MIN_COS, rows[:5]
print("okdmae", float(l1), r1, float(l3), r3, rel[:2], dirs)
"""
    return code


def case_dmae_tpmcl(dev, golden, ptypes=(2, 3, 4)):
    """DmaeUtils.get_partial_similarity (TPM-CL margin losses, types 2 / 3 / 4) on the device vs the reference run."""
    from antmmf.common.configuration import Configuration

    g = golden("ops_dmae_tpmcl.pt")
    mod = load_dmae_utils()
    res = {}
    for ptype in ptypes:
        du = mod.DmaeUtils(Configuration(dict(DMAE_CFG, l3_interaction="wti", l3_with_nfc=True, l3_sim_header="meanP", l3_partial_type=ptype,
                                              l3_max_frames=4, l3_max_words=12)))
        W.fill_module_(du)
        du.tis_selector.thresh.fill_(0.6)
        du = du.to(dev).train()
        t, w_, v = (g[k].detach().clone().to(dev).requires_grad_(True) for k in ("text", "word", "video"))
        loss = du.get_partial_similarity((t, w_), v, g["word_mask"].to(dev), g["video_mask"].to(dev), ptype)
        ref = float(g[f"p{ptype}.loss"])
        assert abs(float(loss) - ref) <= 1e-3 * abs(ref), (ptype, float(loss), ref)
        loss.backward()
        for nm, x in (("dtext", t), ("dword", w_), ("dvideo", v)):
            gr = x.grad if x.grad is not None else torch.zeros_like(x)
            rn = float(g[f"p{ptype}.{nm}.norm"])
            assert abs(float(gr.norm()) - rn) <= 5e-3 * max(rn, 1e-6), (ptype, nm, float(gr.norm()), rn)
        named = dict(du.named_parameters())
        for k in g:
            if k.startswith(f"p{ptype}.gnorm."):
                n = k[len(f"p{ptype}.gnorm."):]
                gn = float(named[n].grad.norm()) if named[n].grad is not None else 0.0
                assert abs(gn - float(g[k])) <= 5e-3 * max(float(g[k]), 1e-5), (ptype, n, gn, float(g[k]))
        res[ptype] = (float(loss), ref)
    return res


def case_dmae_tpmcl_blocks(dev, sim_header="meanP"):
    """TPM-CL over SEVERAL 8 x 16 caption x video blocks: the batched evaluation of all blocks at once (product path at bench sizes: token-importance
    weights of all pairs from per-caption / per-video pieces, LinearXWeightPredictor.forward_all_pairs) equals the reference-shaped Python double
    loop over the blocks with pair-batch predictors (`l3_partial_loop`; the loop form is what ops_dmae_tpmcl.pt pins on one block):
    loss, input gradients, parameter gradients."""
    from antmmf.common.configuration import Configuration

    mod = load_dmae_utils()
    Bt, Bv, Nw, V, D = 16, 32, 12, 4, 128
    out = {}
    for loop in (True, False):
        du = mod.DmaeUtils(Configuration(dict(DMAE_CFG, l3_interaction="wti", l3_with_nfc=True, l3_sim_header=sim_header, l3_partial_type=4,
                                              l3_max_frames=V, l3_max_words=Nw, l3_partial_loop=loop)))
        W.fill_module_(du)
        du.tis_selector.thresh.fill_(0.6)
        du = du.to(dev).train()
        norm = lambda x: x / x.norm(dim=-1, keepdim=True)  # noqa: E731
        t = norm(W.data_tensor("tpmb.text", (Bt, 1, D))).to(dev).requires_grad_(True)
        w_ = norm(W.data_tensor("tpmb.word", (Bt, Nw, D))).to(dev).requires_grad_(True)
        v = norm(W.data_tensor("tpmb.video", (Bv, V + 1, D))).to(dev).requires_grad_(True)
        wm = torch.ones(Bt, Nw, device=dev)
        wm[3, 7:] = 0
        wm[10, 4:] = 0
        vm = torch.ones(Bv, V + 1, device=dev)
        loss = du.get_partial_similarity((t, w_), v, wm, vm, 4)
        loss.backward()
        out[loop] = (float(loss), t.grad.clone(), w_.grad.clone(), v.grad.clone(), {n: p.grad.clone() for n, p in du.named_parameters() if p.grad is not None})
    (l0, dt0, dw0, dv0, g0), (l1, dt1, dw1, dv1, g1) = out[True], out[False]
    assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
    for nm, a, b in (("dtext", dt0, dt1), ("dword", dw0, dw1), ("dvideo", dv0, dv1)):
        torch.testing.assert_close(b, a, rtol=2e-3, atol=1e-6 + 2e-3 * float(a.abs().max()), msg=nm)
    assert set(g0) == set(g1)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], rtol=5e-3, atol=1e-6 + 5e-3 * float(g0[n].abs().max()), msg=n)
    # the weight predictors' 2D -> D/2 Linear on the bf16 MFMA GEMM (what the product does from 8192 pair-tokens on; forced here): same loss to 1e-3,
    # every gradient within bf16 distance of the fp32-matmul run
    du = mod.DmaeUtils(Configuration(dict(DMAE_CFG, l3_interaction="wti", l3_with_nfc=True, l3_sim_header=sim_header, l3_partial_type=4,
                                          l3_max_frames=V, l3_max_words=Nw, l3_xwp_pair_batch=True)))   # pair-batch predictors (not the all-pairs algebra)
    W.fill_module_(du)
    du.tis_selector.thresh.fill_(0.6)
    du = du.to(dev).train()
    saved = mod.LinearXWeightPredictor.MFMA_MIN_ROWS
    mod.LinearXWeightPredictor.MFMA_MIN_ROWS = 0
    try:
        t = norm(W.data_tensor("tpmb.text", (Bt, 1, D))).to(dev).requires_grad_(True)
        w_ = norm(W.data_tensor("tpmb.word", (Bt, Nw, D))).to(dev).requires_grad_(True)
        v = norm(W.data_tensor("tpmb.video", (Bv, V + 1, D))).to(dev).requires_grad_(True)
        loss = du.get_partial_similarity((t, w_), v, wm, vm, 4)
        loss.backward()
    finally:
        mod.LinearXWeightPredictor.MFMA_MIN_ROWS = saved
    assert abs(float(loss) - l1) <= 1e-3 * abs(l1), (float(loss), l1)
    g2 = {n: p.grad for n, p in du.named_parameters() if p.grad is not None}
    assert set(g2) == set(g1)
    worst = 1.0
    top = max(float(x.float().norm()) for x in g1.values())
    for nm, a, b in [("dtext", dt1, t.grad), ("dword", dw1, w_.grad), ("dvideo", dv1, v.grad)] + [(n, g1[n], g2[n]) for n in g1]:
        a, b = a.float().flatten(), b.float().flatten()
        if float(a.norm()) < 1e-4 * top:   # e.g. the weight heads' biases: zero by the shift invariance of the masked softmax
            assert float(b.norm()) < 1e-3 * top, (nm, float(b.norm()))
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        worst = min(worst, cos)
        assert cos >= 0.99 and abs(float(b.norm()) - float(a.norm())) <= 0.05 * float(a.norm()), (nm, cos, float(a.norm()), float(b.norm()))
    return dict(loss=(l0, l1, float(loss)), params=len(g0), bf16_predictor_min_cos=worst)


def case_dmae_wti(dev, golden):
    """DmaeUtils.wti_interaction on the HIP path (split GEMM + fused reduction kernel) vs the reference run: wti / att_wti, with and
    without the second-best-frame term, forward + gradients of the features and of the weight heads."""
    from antmmf.common.configuration import Configuration

    g = golden("ops_dmae_wti.pt")
    mod = load_dmae_utils()
    res = {}
    for inter in ("wti", "att_wti"):
        for va in (True, False):
            tag = f"{inter}.va{int(va)}"
            du = mod.DmaeUtils(Configuration(dict(DMAE_CFG, l3_interaction=inter, l3_with_nfc=va, l3_sim_header="meanP")))
            W.fill_module_(du)
            du = du.to(dev).train()
            t, w_, v = (g[k].to(dev).clone().requires_grad_(True) for k in ("text", "word", "video"))
            out = du.wti_interaction(t, w_, v, g["word_mask"].to(dev), g["video_mask"].to(dev))
            check(f"{tag}.out", out, g[f"{tag}.out"], 1e-3, 1e-3)
            (out * g["g"].to(dev)).sum().backward()
            check(f"{tag}.dtext", t.grad, g[f"{tag}.dtext"], 2e-2, 2e-2)    # bf16 slab gradient through the two GEMMs back
            check(f"{tag}.dvideo", v.grad, g[f"{tag}.dvideo"], 2e-2, 2e-2)
            if inter == "att_wti":
                check(f"{tag}.dword", w_.grad, g[f"{tag}.dword"], 2e-2, 2e-2)
            for n, p in du.named_parameters():
                if f"{tag}.grad.{n}" in g and p.grad is not None:
                    if float(g[f"{tag}.grad.{n}"].abs().max()) < 1e-6:   # bias of a softmax-fed Linear: analytically zero gradient
                        assert float(p.grad.abs().max()) < 1e-5
                        continue
                    check(f"{tag}.grad.{n}", p.grad, g[f"{tag}.grad.{n}"], 1e-2, 1e-2)
            res[tag] = float(out.sum())
    return res


# ------------------------------------------------------------------------------ M2
M2_PRJ = os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "M2_Encoder")
if M2_PRJ not in sys.path:
    sys.path.insert(0, M2_PRJ)

TINY_M2 = dict(beit_version="base", encoder_embed_dim=128, out_embed_dim=64, encoder_layers=2, beit3_vl_layers=1,
               image_size=32, patch_size=8, vocab_size=300, max_text_len=12, encoder_attention_heads=2)


def build_tiny_m2(dev):
    from vlmo.config import default_config
    from vlmo.modules.vlmo_module import VLMo

    cfg = default_config()
    cfg.update(TINY_M2)
    model = VLMo(cfg)
    W.fill_module_(model)
    return model.to(dev).train()


def case_m2_towers(dev, golden, rtol=5e-2):
    """Product VLMo.infer_image / infer_text + logits (bf16 HIP path) vs the reference's outputs, and gradients of the
    reference-pinned scalar `pin` (make_golden.gen_e2e_m2)."""
    g = golden("e2e_m2.pt")
    model = build_tiny_m2(dev)
    oi = model.infer_image({"image": [g["image"].to(dev)]})
    ot = model.infer_text({"text_ids": g["text_ids"].to(dev), "text_masks": g["text_masks"].to(dev)})
    check("m2.image_feats", oi["image_feats"], g["img.image_feats"], rtol, 5e-2)
    for k in ("cls_feats", "cls_vlffn_feats"):
        check(f"m2.img.{k}", oi[k], g[f"img.{k}"], rtol, 4e-2)
        check(f"m2.txt.{k}", ot[k], g[f"txt.{k}"], rtol, 4e-2)
    logits = model.logit_scale.exp() * oi["cls_feats"] @ ot["cls_feats"].t()
    logits_vl = model.logit_vl_scale.exp() * oi["cls_vlffn_feats"] @ ot["cls_vlffn_feats"].t()
    check("m2.logits", logits, g["logits"], rtol, 4e-2)
    pin = (logits * W.data_tensor("m2.wl", (3, 3)).to(dev)).sum() + (logits_vl * W.data_tensor("m2.wvl", (3, 3)).to(dev)).sum()
    from antmmf.hip import functional as HF
    handoffs0 = HF.COLSUM_HANDOFFS[0]
    pin.backward()
    # stacked layers: every fc2 bias gradient but the last layer's comes from the next layer's ln1 backward (no column-sum pass); the
    # gradient checks below (all parameters, fc2 biases included) are what proves the handed-over sums are the right ones
    assert HF.COLSUM_HANDOFFS[0] - handoffs0 >= 2, HF.COLSUM_HANDOFFS[0] - handoffs0
    worst = []
    for n, p in model.named_parameters():
        if f"gnorm.{n}" not in g:
            continue
        assert p.grad is not None, f"no grad for {n}"
        worst.append([abs(float(p.grad.float().norm()) - float(g[f"gnorm.{n}"])), n, float(p.grad.float().norm()), float(g[f"gnorm.{n}"])])
        if f"grad.{n}" in g:
            check(f"m2.grad.{n}", p.grad, g[f"grad.{n}"], 1e-1, 1e-1)
    top = max(w[3] for w in worst)
    kept = []
    for w in worst:
        if w[3] < 1e-6 * top:
            assert w[2] < 1e-3 * top, f"{w[1]}: gradient should vanish, got norm {w[2]}"
            continue
        w[0] /= w[3]
        kept.append(w)
    kept.sort(reverse=True)
    assert len(kept) > 50 and kept[0][0] < 0.05, f"gradient norms off: {kept[:5]}"
    dirs = assert_grad_directions(model.named_parameters(), g, "", min_checked=50)
    return dict(pin=float(pin), ref_pin=float(g["pin"]), worst_gnorm=kept[:3], directions=dirs)


def case_m2_itc_vs_oracle(dev, loss_batches=1):
    """Product VLMo training step (sharded ITC, world 1) vs the CPU oracle's M2 ITC step on the same weights.
    loss_batches > 1: the 1e-3 loss contract is checked on the MEAN over that many seeded batches instead of on the first one alone.  A 4-pair InfoNCE
    loss at logit scale 14 turns the ~1 % bf16 error of the embeddings into a per-batch deviation of sigma ~ 1e-3 whatever the kernels (measured on five
    batches: separate LayerNorm kernels +4.7e-4, -1.0e-3, -2.0e-3, +9.7e-4, +3.8e-4; sub-LN fold -3.5e-4, -6.4e-4, -1.9e-3, +9.8e-4, +1.1e-3; tower
    outputs against the reference goldens 0.99 % vs 0.94 %), so a single tiny batch sits AT the contract's noise floor; gradients are checked on the first batch."""
    from oracle import step as ostep

    model = build_tiny_m2(dev)
    lengths = torch.tensor([12, 5, 8, 3])
    mask = (torch.arange(12)[None, :] < lengths[:, None]).long()

    def batch(tag):
        im = (W.data_tensor(f"m2s.image{tag}", (4, 3, 32, 32)) * 0.25 + 0.5).clamp(0, 1)
        return im, W.data_ints(f"m2s.ids{tag}", (4, 12), 1, 300) * mask

    img, ids = batch("")
    out = model({"image": [img.to(dev)], "text_ids": ids.to(dev), "text_masks": mask.to(dev)})
    loss = out["losses"]["itc_loss"] + out["losses"]["itc_vl_loss"]
    P = tiny_models.m2_params(requires_grad=True)
    ref = ostep.m2_itc(P, img, ids, mask, heads=2, patch=8)
    rel = [(float(loss) - float(ref["loss"])) / abs(float(ref["loss"]))]
    with torch.no_grad():
        for k in range(1, loss_batches):
            im_k, ids_k = batch(f".{k}")
            o_k = model({"image": [im_k.to(dev)], "text_ids": ids_k.to(dev), "text_masks": mask.to(dev)})
            r_k = float(ostep.m2_itc(P, im_k, ids_k, mask, heads=2, patch=8)["loss"])
            rel.append((float(o_k["losses"]["itc_loss"] + o_k["losses"]["itc_vl_loss"]) - r_k) / abs(r_k))
    assert abs(sum(rel) / len(rel)) <= 1e-3 and max(abs(r) for r in rel) <= (1e-3 if loss_batches == 1 else 5e-3), rel   # north_star: loss within 1e-3 rel
    loss.backward()
    ref["loss"].backward()
    named = dict(model.named_parameters())
    rels = []
    gold = {f"gfull.{n}": p.grad for n, p in P.items() if p.grad is not None and n in named}
    for n, p in P.items():
        if p.grad is None or n not in named or named[n].grad is None:
            continue
        rn = float(p.grad.norm())
        if rn < 1e-6:
            continue
        rels.append((abs(float(named[n].grad.float().norm()) - rn) / rn, n))
    rels.sort(reverse=True)
    assert len(rels) > 50 and rels[0][0] < 0.05, rels[:5]
    dirs = assert_grad_directions(named.items(), gold, "", min_checked=50)
    return dict(loss=float(loss), ref=float(ref["loss"]), worst=rels[:3], directions=dirs)


# ------------------------------------------------------------------------------ one transformer layer at REAL width
def case_layer_real_width(dev, kind="m2", d=1024, heads=16, N=257, B=2, pad_tail=0):
    """`transformer_layer` forward + backward at the widths of BASELINE.json's configs (M2 ViT-L/14: d = 1024, 16 heads, 257 tokens; CLIP-arch
    ViT-B/16: d = 768, 12 heads, 197 tokens) against the fp32 oracle layer on the same bf16-rounded weights and input: output element-wise,
    input gradient and EVERY parameter gradient by cosine (>= 0.999) and norm (2 %).  The tiny-dim fixtures cannot see an indexing error that only
    shows up at 16 heads x 64 or past the first 128 / 256-wide tile; this can."""
    from antmmf.hip import functional as HF
    from kernel_cases import q, rnd
    from oracle import towers as otowers

    g = torch.Generator().manual_seed(1234 + d + N)

    def w(shape, scale):
        return q(torch.randn(shape, generator=g) * scale)

    P = {}
    if kind == "m2":
        spec = HF.LayerSpec(kind="m2", heads=heads, eps=1e-5, act="gelu", packed_qkv=False)
        names = dict(ln1="self_attn_layer_norm.A", inner="self_attn.inner_attn_ln.A", ln2="final_layer_norm.A", ffn="ffn.A.ffn_layernorm")
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            P[f"self_attn.{nm}.A.weight"], P[f"self_attn.{nm}.A.bias"] = w((d, d), d ** -0.5), w((d,), 0.1)
        P["ffn.A.fc1.weight"], P["ffn.A.fc1.bias"] = w((4 * d, d), d ** -0.5), w((4 * d,), 0.1)
        P["ffn.A.fc2.weight"], P["ffn.A.fc2.bias"] = w((d, 4 * d), (4 * d) ** -0.5), w((d,), 0.1)
        for key, width in ((names["ln1"], d), (names["inner"], d), (names["ln2"], d), (names["ffn"], 4 * d)):
            P[key + ".weight"], P[key + ".bias"] = q(1.0 + 0.1 * torch.randn(width, generator=g)), w((width,), 0.1)
        slots = dict(ln1_w=names["ln1"] + ".weight", ln1_b=names["ln1"] + ".bias", wq="self_attn.q_proj.A.weight", bq="self_attn.q_proj.A.bias",
                     wk="self_attn.k_proj.A.weight", bk="self_attn.k_proj.A.bias", wv="self_attn.v_proj.A.weight", bv="self_attn.v_proj.A.bias",
                     inner_w=names["inner"] + ".weight", inner_b=names["inner"] + ".bias", wo="self_attn.out_proj.A.weight", bo="self_attn.out_proj.A.bias",
                     ln2_w=names["ln2"] + ".weight", ln2_b=names["ln2"] + ".bias", w1="ffn.A.fc1.weight", b1="ffn.A.fc1.bias",
                     ffn_w=names["ffn"] + ".weight", ffn_b=names["ffn"] + ".bias", w2="ffn.A.fc2.weight", b2="ffn.A.fc2.bias")
    else:
        spec = HF.LayerSpec(kind="clip", heads=heads, eps=1e-5, act="quick_gelu", packed_qkv=True)
        P["attn.in_proj_weight"], P["attn.in_proj_bias"] = w((3 * d, d), d ** -0.5), w((3 * d,), 0.1)
        P["attn.out_proj.weight"], P["attn.out_proj.bias"] = w((d, d), d ** -0.5), w((d,), 0.1)
        P["mlp.c_fc.weight"], P["mlp.c_fc.bias"] = w((4 * d, d), d ** -0.5), w((4 * d,), 0.1)
        P["mlp.c_proj.weight"], P["mlp.c_proj.bias"] = w((d, 4 * d), (4 * d) ** -0.5), w((d,), 0.1)
        for ln in ("ln_1", "ln_2"):
            P[ln + ".weight"], P[ln + ".bias"] = q(1.0 + 0.1 * torch.randn(d, generator=g)), w((d,), 0.1)
        slots = dict(ln1_w="ln_1.weight", ln1_b="ln_1.bias", wqkv="attn.in_proj_weight", bqkv="attn.in_proj_bias", wo="attn.out_proj.weight",
                     bo="attn.out_proj.bias", ln2_w="ln_2.weight", ln2_b="ln_2.bias", w1="mlp.c_fc.weight", b1="mlp.c_fc.bias",
                     w2="mlp.c_proj.weight", b2="mlp.c_proj.bias")
    x = w((B, N, d), 1.0)
    G = w((B, N, d), 1.0)
    pad = None
    if pad_tail and kind == "m2":
        pad = torch.zeros(B, N, dtype=torch.bool)
        pad[1, N - pad_tail:] = True
    # oracle (fp32, CPU)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    yr = otowers.m2_layer(Pr, xr, "A", heads, pad=pad) if kind == "m2" else otowers.clip_block(Pr, xr, heads)
    keep = torch.ones(B, N, 1) if pad is None else (~pad).float()[..., None]
    (yr * G * keep).sum().backward()
    # product
    Pd = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in P.items()}
    xd = x.to(dev, torch.bfloat16).requires_grad_(True)
    key_bias = None if pad is None else torch.zeros(B, N, dtype=torch.float32, device=dev).masked_fill_(pad.to(dev), float("-inf"))
    yd = HF.transformer_layer(xd, spec, {s: Pd[k] for s, k in slots.items()}, key_bias)
    (yd.float() * (G * keep).to(dev)).sum().backward()
    sel = keep.bool().expand_as(yr)
    check(f"layer.{kind}.{d}.y", torch.where(sel.to(dev), yd.float(), torch.zeros_like(yd.float())), torch.where(sel, yr, torch.zeros_like(yr)), 2e-2, 1e-2)
    rows = []

    def cmp(name, got, ref):
        got, ref = got.detach().float().flatten().cpu(), ref.detach().float().flatten()
        gn, rn = float(got.norm()), float(ref.norm())
        rows.append((float(torch.dot(got, ref)) / max(gn * rn, 1e-30), abs(gn - rn) / max(rn, 1e-30), name))

    cmp("dx", xd.grad, xr.grad)
    top = max(float(Pr[k].grad.norm()) for k in P)
    for k in P:
        if float(Pr[k].grad.norm()) < 1e-5 * top:   # the key-projection bias: zero by softmax shift invariance (fp32 rounding noise in the oracle)
            assert float(Pd[k].grad.float().norm()) < 1e-2 * top, (k, float(Pd[k].grad.float().norm()), top)
            continue
        cmp(k, Pd[k].grad, Pr[k].grad)
    rows.sort()
    assert rows[0][0] >= 0.999, f"gradient direction off: {rows[:4]}"
    worst = max(rows, key=lambda r: r[1])
    assert worst[1] <= 0.02, f"gradient norm off: {worst}"
    return dict(min_cos=rows[0], worst_norm=worst, n=len(rows))


def case_bert_layer_cls_only_gradient(dev, d=768, heads=12, N=77, B=8):
    """The LAST BertLayer of a text tower as the contrastive step drives it: only the [CLS] row of its output feeds the loss (pooling = token 0), so the gradient that
    reaches the layer is zero on every other token, and its query / key projections see one query per sequence -- their weight gradients are ~ 1e-4 of the layer's largest.
    In the whole-model real-width cases those two come out at cosine 0.02 ... 0.3 against the oracle ("plausible, but nothing pins them"): there the layer's INPUT
    already carries eleven layers of bf16 noise.  Here the layer gets the same bf16-rounded input, weights and [CLS]-only upstream gradient as the fp32 oracle layer, ragged
    -10000 key masks as in training: every parameter gradient, the tiny ones included, must agree in direction (>= 0.995) and norm (3 %) -- the kernels are right for them;
    what the end-to-end cosine shows is the upstream noise."""
    from antmmf.hip import functional as HF
    from kernel_cases import q
    from oracle import towers as otowers

    g = torch.Generator().manual_seed(4321)

    def w(shape, scale):
        return q(torch.randn(shape, generator=g) * scale)

    P = {}
    for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
        P[nm + ".weight"], P[nm + ".bias"] = w((d, d), d ** -0.5), w((d,), 0.1)
    P["intermediate.dense.weight"], P["intermediate.dense.bias"] = w((4 * d, d), d ** -0.5), w((4 * d,), 0.1)
    P["output.dense.weight"], P["output.dense.bias"] = w((d, 4 * d), (4 * d) ** -0.5), w((d,), 0.1)
    for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
        P[ln + ".weight"], P[ln + ".bias"] = q(1.0 + 0.1 * torch.randn(d, generator=g)), w((d,), 0.1)
    slots = dict(wq="attention.self.query.weight", bq="attention.self.query.bias", wk="attention.self.key.weight", bk="attention.self.key.bias",
                 wv="attention.self.value.weight", bv="attention.self.value.bias", wo="attention.output.dense.weight", bo="attention.output.dense.bias",
                 ln1_w="attention.output.LayerNorm.weight", ln1_b="attention.output.LayerNorm.bias", w1="intermediate.dense.weight", b1="intermediate.dense.bias",
                 w2="output.dense.weight", b2="output.dense.bias", ln2_w="output.LayerNorm.weight", ln2_b="output.LayerNorm.bias")
    x = w((B, N, d), 1.0)
    lengths = torch.randint(5, N + 1, (B,), generator=g)
    lengths[0] = N
    key_bias = torch.zeros(B, N).masked_fill(torch.arange(N)[None, :] >= lengths[:, None], -10000.0)
    G = torch.zeros(B, N, d)
    G[:, 0] = w((B, d), 1.0)          # the pooled [CLS] row only
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    yr = otowers.bert_layer(Pr, xr, key_bias, heads)
    (yr * G).sum().backward()
    spec = HF.LayerSpec(kind="bert", heads=heads, eps=1e-12, act="gelu", packed_qkv=False)
    Pd = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in P.items()}
    xd = x.to(dev, torch.bfloat16).requires_grad_(True)
    yd = HF.transformer_layer(xd, spec, {s_: Pd[k] for s_, k in slots.items()}, key_bias.to(dev))
    (yd.float() * G.to(dev)).sum().backward()
    check("bert_cls.y0", yd[:, 0].float(), yr[:, 0], 2e-2, 1e-2)
    top = max(float(Pr[k].grad.norm()) for k in P)
    rows = []
    for k in list(P) + ["dx"]:
        ref = (xr.grad if k == "dx" else Pr[k].grad).detach().float().flatten()
        got = (xd.grad if k == "dx" else Pd[k].grad).detach().float().flatten().cpu()
        rn, gn = float(ref.norm()), float(got.norm())
        if rn < 1e-6 * top:   # the key bias: zero by softmax shift invariance
            assert gn < 1e-3 * top, (k, gn, top)
            continue
        rows.append((float(torch.dot(got, ref)) / max(gn * rn, 1e-30), abs(gn - rn) / rn, k, rn / top))
    rows.sort()
    small = [r for r in rows if r[3] < 1e-2]
    assert any(r[2] == "attention.self.query.weight" for r in rows) and any(r[2] == "attention.self.key.weight" for r in rows)
    assert rows[0][0] >= 0.995, f"gradient direction off: {rows[:4]}"
    assert max(r[1] for r in rows) <= 0.03, f"gradient norm off: {max(rows, key=lambda r: r[1])}"
    return dict(min_cos=rows[0], n=len(rows), small_share_parameters=[(r[2], round(r[3], 6), round(r[0], 5)) for r in small])


# ------------------------------------------------------------------------------ ViLBERT co-attention operator (T12)
def case_vilbert_biattention(dev, golden, head_size=64):
    """BertBiAttention (antmmf/models/vilbert.py:285-416) on the HIP path vs the reference run (ops_vilbert_biattention.pt: both contexts, input
    gradients, every parameter gradient; ragged key masks on both streams), and -- training mode, dropout on -- vs the oracle co-attention with
    the same counter-based masks rebuilt on the host."""
    import numpy as np
    from antmmf.common.configuration import Configuration
    from antmmf.models.vilbert import BertBiAttention
    from kernel_cases import dropout_keep_np
    from oracle import ops as oops

    g_all = golden("ops_vilbert_biattention.pt")
    prefix = "" if head_size == 64 else f"h{head_size}."
    g = {k[len(prefix):]: v for k, v in g_all.items() if k.startswith(prefix) and (prefix or not k.startswith("h128."))}
    cfg = Configuration(dict(bi_hidden_size=2 * head_size, bi_num_attention_heads=2, v_hidden_size=96, hidden_size=128, v_attention_probs_dropout_prob=0.1,
                             attention_probs_dropout_prob=0.1, visualization=False))
    m = BertBiAttention(cfg)
    W.fill_module_(m)
    m = m.to(dev).eval()
    x1 = g["x1"].to(dev).requires_grad_(True)
    x2 = g["x2"].to(dev).requires_grad_(True)
    c1, c2, vis = m(x1, g["mask1"].to(dev), x2, g["mask2"].to(dev))
    assert vis is None and c1.shape == g["ctx1"].shape and c2.shape == g["ctx2"].shape
    check("bi.ctx1", c1, g["ctx1"], 3e-2, 2e-2)
    check("bi.ctx2", c2, g["ctx2"], 3e-2, 2e-2)
    ((c1.float() * g["g1"].to(dev)).sum() + (c2.float() * g["g2"].to(dev)).sum()).backward()
    check("bi.dx1", x1.grad, g["dx1"], 5e-2, 3e-2)
    check("bi.dx2", x2.grad, g["dx2"], 5e-2, 3e-2)
    rows = []
    for n, p in m.named_parameters():
        ref = g["grad." + n].float().flatten()
        got = p.grad.detach().float().flatten().cpu()
        if float(ref.norm()) < 1e-5:      # key biases: zero by shift invariance
            continue
        rows.append((float(torch.dot(got, ref)) / float(got.norm() * ref.norm()), abs(float(got.norm()) - float(ref.norm())) / float(ref.norm()), n))
    rows.sort()
    assert len(rows) >= 10 and rows[0][0] >= 0.995 and max(r[1] for r in rows) <= 0.05, rows[:4]
    # training mode: attention-probability dropout inside the kernels, masks = keep(seed, ((b h + head) Nq + q) Nk + k)
    m.train()
    seeds = [(5 << 32) | 1234, (9 << 32) | 77]
    with torch.no_grad():
        d1, d2, _ = m(g["x1"].to(dev), g["mask1"].to(dev), g["x2"].to(dev), g["mask2"].to(dev), dropout_seeds=seeds)
    P = {n: p.detach().float().cpu() for n, p in m.named_parameters()}
    xa, xb = g["x1"].to(torch.bfloat16).float(), g["x2"].to(torch.bfloat16).float()

    def proj(x, s):
        return [oops.split_heads((x @ P[f"{k}{s}.weight"].to(torch.bfloat16).float().t() + P[f"{k}{s}.bias"]).to(torch.bfloat16).float(), 2)
                for k in ("query", "key", "value")]

    (q1, k1, v1), (q2, k2, v2) = proj(xa, 1), proj(xb, 2)
    B, Nv, Nt = xa.shape[0], xa.shape[1], xb.shape[1]
    for tag, got, qq, kk, vv, bias, seed, nq, nk in (("ctx1", d1, q2, k1, v1, g["mask1"].reshape(B, -1), seeds[0], Nt, Nv),
                                                    ("ctx2", d2, q1, k2, v2, g["mask2"].reshape(B, -1), seeds[1], Nv, Nt)):
        keep = dropout_keep_np(np.arange(B * 2 * nq * nk).reshape(B, 2, nq, nk), seed, 0.1).float() / 0.9
        s = torch.matmul(qq, kk.transpose(-1, -2)) / math.sqrt(head_size) + bias[:, None, None, :]
        want = oops.merge_heads(torch.matmul(torch.softmax(s, -1) * keep, vv))
        check("bi.drop." + tag, got, want, 3e-2, 2e-2)
    with pytest_raises(NotImplementedError):   # head sizes other than 64 / 128 are refused, not silently mis-computed
        BertBiAttention(Configuration(dict(cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg), bi_hidden_size=192)))
    return dict(min_cos=rows[0], n=len(rows))


def pytest_raises(exc):
    import pytest

    return pytest.raises(exc)
