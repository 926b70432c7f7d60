#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by EXECUTING the reference implementation.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference modules are imported unmodified through `_ref_loader`;
parameters are filled by `weightgen` (name-keyed, deterministic) so that the fixtures hold only
inputs and expected outputs / gradients.  Each fixture is a flat dict of float32/int64 tensors saved
with torch.save (a few KB each).

Fixtures and the reference symbols that produced them:
  ops_clip_block.pt    ResidualAttentionBlock            antmmf/modules/vision/backbone/clip/model.py:227-256
  ops_bert_layer.pt    BertLayer (+additive -10000 mask)  antmmf/modules/vision/backbone/clip/modeling_bert.py:253-270
  ops_m2_layer.pt      torchscale EncoderLayer A/B branch prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:113-168
  loss_mil_nce.pt      get_mil_nce_loss                   prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:146-197
  loss_misc.pt         moco_loss / CrossEn / NegNCE       moco_utils.py:71-81, prj/dmae_vtp/.../dmae_utils.py:528-563
  ops_dmae_seqtransf.pt DmaeUtils._agg_visual_feat(seqTransf)  prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:186-227,574-619
  ops_dmae_tpmcl.pt    DmaeUtils.get_partial_similarity (TPM-CL margin losses, partial types 2 / 3 / 4)   dmae_utils.py:280-523, tpmcl_utils.py
  ops_dmae_wti.pt      DmaeUtils.wti_interaction (wti / att_wti, with and without the 2nd-frame term)   dmae_utils.py:85-184
  metric_recall.pt     _cal_recall / _cal_sym_recall (retrieval evaluation)   antmmf/modules/metrics/global_retrieval_recall.py:13-103
  m2_ckpt_convert.pt   convert_pl_ckpt / convert_deepspeed_ckpt (position-table resize)   prj/M2_Encoder/vlmo/modules/vlmo_module.py:22-106
  e2e_clip_arch.pt     UnivlForVideoTextRetrieval stage1  univl_video_ret.py:357-387,457-480 (tiny ViT + tiny BERT)
  e2e_clip_stage2.pt   same model, training_stage stage1+stage2 (cross encoder; plain and hard-mining+median reweight); the
                       reference instance's nn.Dropout(0.1) in front of similarity_dense is set to p = 0   univl_video_ret.py:33-144,389-443
  e2e_cnvid_gate.pt    cnvid_vtp UnivlForVideoTextRetrieval.forward_stage(incre_num): mined / plain decision per (seed, incre_num)   prj/cnvid_vtp/.../univl_video_ret.py:398-428
  e2e_dmae_stage3.pt   dmae_vtp UnivlForVideoTextRetrieval, stage1+stage3 (seqTransf + WTI + NegNCE / CrossEn)   prj/dmae_vtp/.../univl_video_ret.py:457-476
  e2e_clip_moco.pt     same model, with_moco: true (K=64, M=0.5): 2 steps   univl_video_ret.py:262-312, moco_utils.py:13-107
  e2e_m2.pt            VLMo.infer_image / infer_text      prj/M2_Encoder/vlmo/modules/vlmo_module.py:323-405 (tiny dims)
  gather_w2.pt         gather_tensor(back_gradient=True)  antmmf/utils/distributed_utils.py:92-189 (2-proc gloo)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402
import weightgen as W  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)


def save(name, d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().clone().contiguous()
        out[k] = v
    path = os.path.join(HERE, name)
    torch.save(out, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} entries")


def grads_of(module, prefix="grad."):
    return {prefix + n: p.grad for n, p in module.named_parameters() if p.grad is not None}


# ----------------------------------------------------------------------------- per-op fixtures
def gen_clip_block():
    vit = L.load_antmmf_core()["vit"]
    blk = vit.ResidualAttentionBlock(128, 2)
    W.fill_module_(blk)
    x = W.data_tensor("clip_block.x", (17, 3, 128)).requires_grad_(True)  # LND
    w = W.data_tensor("clip_block.w", (17, 3, 128))
    y = blk(x)
    (y * w).sum().backward()
    d = dict(x=x, w=w, y=y, dx=x.grad)
    d.update(grads_of(blk))
    save("ops_clip_block.pt", d)


def gen_bert_layer():
    core = L.load_antmmf_core()
    cfg = core["bert_cfg"].BertConfig(
        vocab_size_or_config_json_file=100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
        intermediate_size=512, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        layer_norm_eps=1e-12)
    layer = core["bert"].BertLayer(cfg)
    W.fill_module_(layer)
    x = W.data_tensor("bert_layer.x", (3, 12, 128)).requires_grad_(True)
    w = W.data_tensor("bert_layer.w", (3, 12, 128))
    lengths = torch.tensor([12, 7, 3])
    mask = (torch.arange(12)[None, :] < lengths[:, None]).long()
    ext = (1.0 - mask[:, None, None, :].float()) * -10000.0
    y = layer(x, ext, None)
    (y * w).sum().backward()
    d = dict(x=x, w=w, mask=mask, y=y, dx=x.grad)
    d.update(grads_of(layer))
    save("ops_bert_layer.pt", d)


def gen_m2_layer():
    L.load_m2()
    from vlmo.torchscale.architecture.config import EncoderConfig
    from vlmo.torchscale.architecture.encoder import EncoderLayer

    args = EncoderConfig(multiway=True, encoder_embed_dim=128, encoder_attention_heads=2, encoder_ffn_embed_dim=512,
                         encoder_layers=2, layernorm_embedding=False, normalize_output=True, no_output_layer=True,
                         img_size=32, patch_size=8, vocab_size=100)
    layer = EncoderLayer(args, depth=0)
    W.fill_module_(layer)
    d = {}
    lengths = torch.tensor([12, 7, 3])
    pad = ~(torch.arange(12)[None, :] < lengths[:, None])
    for tag, split, padmask in (("A", -1, None), ("B", 0, pad)):
        layer.zero_grad()
        x = W.data_tensor(f"m2_layer.x{tag}", (3, 12, 128)).requires_grad_(True)
        w = W.data_tensor(f"m2_layer.w{tag}", (3, 12, 128))
        y, _ = layer(x, encoder_padding_mask=padmask, multiway_split_position=split)
        (y * w).sum().backward()
        d.update({f"{tag}.x": x, f"{tag}.w": w, f"{tag}.y": y, f"{tag}.dx": x.grad})
        d.update(grads_of(layer, prefix=f"{tag}.grad."))
    d["pad"] = pad
    save("ops_m2_layer.pt", d)


DMAE_CFG = dict(hidden_size=128, l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="seqTransf", l3_partial_type=-1,
                l3_max_frames=6, l3_max_words=12, l3_sim_header_hidden_layer=2)


def gen_dmae_seqtransf():
    vtp = L.load_vtp("dmae_vtp")
    du = vtp["dmae"].DmaeUtils(L.AttrDict(DMAE_CFG))
    W.fill_module_(du)
    v = W.data_tensor("dmae.visual", (3, 6, 128)).requires_grad_(True)
    w = W.data_tensor("dmae.w", (3, 6, 128))
    lengths = torch.tensor([6, 4, 2])
    mask = (torch.arange(6)[None, :] < lengths[:, None]).float()
    out, tok_mask, orig = du._agg_visual_feat(v, mask, "seqTransf")
    (out * w).sum().backward()
    d = dict(visual=v, w=w, mask=mask, out=out, tok_mask=tok_mask, dvisual=v.grad)
    for k, g_ in grads_of(du).items():  # full gradients for the small tensors, norms for the weight matrices (keeps the fixture small)
        if g_.numel() <= 1024:
            d[k] = g_
        else:
            d[k.replace("grad.", "gnorm.", 1)] = g_.norm()
            d[k.replace("grad.", "gprobe.", 1)] = g_.flatten()[:64].clone()
    save("ops_dmae_seqtransf.pt", d)


def gen_dmae_wti():
    vtp = L.load_vtp("dmae_vtp")
    d = {}
    A = B = 4
    Nw, V, D = 6, 5, 128
    nrm = torch.nn.functional.normalize
    text = nrm(W.data_tensor("wti.text", (A, 1, D)), dim=-1)
    word = nrm(W.data_tensor("wti.word", (A, Nw, D)), dim=-1)
    video = nrm(W.data_tensor("wti.video", (B, V, D)), dim=-1)
    wlen, vlen = torch.tensor([6, 4, 2, 5]), torch.tensor([5, 3, 4, 1])
    word_mask = (torch.arange(Nw)[None, :] < wlen[:, None]).float()
    video_mask = (torch.arange(V)[None, :] < vlen[:, None]).float()
    gw = W.data_tensor("wti.g", (A, B))
    d.update(dict(text=text, word=word, video=video, word_mask=word_mask, video_mask=video_mask, g=gw))
    for inter in ("wti", "att_wti"):
        for va in (True, False):
            du = vtp["dmae"].DmaeUtils(L.AttrDict(dict(DMAE_CFG, l3_interaction=inter, l3_with_nfc=va, l3_sim_header="meanP")))
            W.fill_module_(du)
            du.train()
            t, w_, v = (x.clone().requires_grad_(True) for x in (text, word, video))
            out = du.wti_interaction(t, w_, v, word_mask.clone(), video_mask.clone())
            (out * gw).sum().backward()
            tag = f"{inter}.va{int(va)}"
            d.update({f"{tag}.out": out, f"{tag}.dtext": t.grad, f"{tag}.dvideo": v.grad,
                      f"{tag}.dword": w_.grad if w_.grad is not None else torch.zeros_like(word)})
            for k, g_ in grads_of(du).items():
                d[f"{tag}.{k}"] = g_
    save("ops_dmae_wti.pt", d)


def gen_metric_recall():
    import importlib
    import types

    L.load_antmmf_core()
    reg = types.ModuleType("antmmf.common.registry")
    reg.registry = types.SimpleNamespace(register_metric=lambda name: (lambda c: c))
    sys.modules["antmmf.common.registry"] = reg
    L._pkg("antmmf.modules.metrics", f"{L.REF}/antmmf/modules/metrics")
    m = importlib.import_module("antmmf.modules.metrics.global_retrieval_recall")
    d = {}
    sq = W.data_tensor("metric.square", (23, 23))
    for k, v in m._cal_recall(sq).items():
        d["sq." + k] = torch.tensor(float(v), dtype=torch.float64)
    d["sq.sim"] = sq
    T, V = 17, 11                                   # 17 captions, 11 videos, several captions per video
    sim = W.data_tensor("metric.rect", (T, V))
    t2v = [[i % V] for i in range(T)]
    v2t = [[i for i in range(T) if i % V == j] for j in range(V)]
    for k, v in m._cal_sym_recall(sim.numpy(), t2v, v2t).items():
        d["rect." + k] = torch.tensor(float(v), dtype=torch.float64)
    d["rect.sim"] = sim
    save("metric_recall.pt", d)


def gen_m2_ckpt_convert():
    L.load_m2()
    from vlmo.modules import vlmo_module as vm

    dim = 8
    d = {}
    for tag, rows in (("grow", 3 + 16), ("shrink", 3 + 64), ("same", 3 + 36)):
        sd = {"backbone.encoder.embed_positions.A.weight": W.data_tensor(f"ckpt.{tag}.pos", (rows, dim)),
              "visual_tokenizer.x": torch.ones(2), "other.weight": W.data_tensor(f"ckpt.{tag}.o", (3, 2))}
        out = vm.convert_pl_ckpt(dict(sd), num_visual_token=37)
        d[f"pl.{tag}.pos"] = out["backbone.encoder.embed_positions.A.weight"]
        d[f"pl.{tag}.nkeys"] = torch.tensor(len(out))
    ds = {"_forward_module.backbone.encoder.embed_positions.A.weight": W.data_tensor("ckpt.ds.pos", (3 + 16, dim)),
          "_forward_module.visual_tokenizer.encoder.pos_embed": W.data_tensor("ckpt.ds.vt", (1, 1 + 16, dim)),
          "_forward_module.head.weight": W.data_tensor("ckpt.ds.h", (2, 2)), "plain.bias": torch.zeros(2)}
    out = vm.convert_deepspeed_ckpt(dict(ds), num_visual_token=37)
    d["ds.pos"] = out["backbone.encoder.embed_positions.A.weight"]
    d["ds.vt"] = out["visual_tokenizer.encoder.pos_embed"]
    d["ds.keys"] = torch.tensor(sorted(len(k) for k in out))
    save("m2_ckpt_convert.pt", d)


def gen_dmae_tpmcl():
    vtp = L.load_vtp("dmae_vtp")
    d = {}
    B, Nw, V, D = 18, 12, 5, 128
    nrm = torch.nn.functional.normalize
    text = nrm(W.data_tensor("tpm.text", (B, 1, D)), dim=-1)
    word = nrm(W.data_tensor("tpm.word", (B, Nw, D)), dim=-1)
    video = nrm(W.data_tensor("tpm.video", (B, V, D)), dim=-1)
    wlen = W.data_ints("tpm.wlen", (B,), 3, Nw + 1)
    word_mask = (torch.arange(Nw)[None, :] < wlen[:, None]).float()
    video_mask = torch.ones(B, V)
    d.update(dict(text=text, word=word, video=video, word_mask=word_mask, video_mask=video_mask))
    for ptype in (2, 3, 4):
        du = vtp["dmae"].DmaeUtils(L.AttrDict(dict(DMAE_CFG, l3_interaction="wti", l3_with_nfc=True, l3_sim_header="meanP", l3_partial_type=ptype,
                                                   l3_max_frames=V - 1, l3_max_words=Nw)))
        W.fill_module_(du)
        du.tis_selector.thresh.fill_(0.6)   # the name-keyed fill also hits this buffer: restore the configured l3_cis_thresh
        du.train()
        t, w_, v = (x.clone().requires_grad_(True) for x in (text, word, video))
        loss = du.get_partial_similarity((t, w_), v, word_mask.clone(), video_mask.clone(), ptype)
        loss.backward()
        d[f"p{ptype}.loss"] = loss
        for nm, x in (("dtext", t), ("dword", w_), ("dvideo", v)):   # norms + a probe keep the fixture small
            g_ = x.grad if x.grad is not None else torch.zeros_like(x)
            d[f"p{ptype}.{nm}.norm"] = g_.norm()
            d[f"p{ptype}.{nm}.probe"] = g_.flatten()[:256].clone()
        for k, g_ in grads_of(du).items():
            d[f"p{ptype}.{k.replace('grad.', 'gnorm.', 1)}"] = g_.norm()
    save("ops_dmae_tpmcl.pt", d)


def gen_losses():
    vtp = L.load_vtp("base_vtp")
    Ret = vtp["ret"].UnivlForVideoTextRetrieval
    d = {}
    for (b, n, use_w) in [(4, 1, False), (8, 1, False), (6, 2, True), (3, 3, False), (5, 1, True)]:
        s = W.data_tensor(f"milnce.{b}.{n}", (b * n, b * n), scale=1.5).requires_grad_(True)
        wv = (W.data_tensor(f"milnce.w.{b}.{n}", (b,)).abs() + 0.1) if use_w else None
        loss = Ret.get_mil_nce_loss(None, s, b, n, wv)
        loss.backward()
        key = f"b{b}n{n}"
        d.update({f"{key}.sim": s, f"{key}.loss": loss, f"{key}.dsim": s.grad})
        if wv is not None:
            d[f"{key}.weight"] = wv
    save("loss_mil_nce.pt", d)

    d = {}
    moco = vtp["moco"].MocoUtils

    class _T:
        T = 0.05

    pos = W.data_tensor("moco.pos", (6, 2), 0.5).requires_grad_(True)
    neg = W.data_tensor("moco.neg", (6, 40), 0.5).requires_grad_(True)
    loss = moco.moco_loss(_T, pos, neg)
    loss.backward()
    d.update({"moco.pos": pos, "moco.neg": neg, "moco.loss": loss, "moco.dpos": pos.grad, "moco.dneg": neg.grad})
    dm = L.load_vtp("dmae_vtp")["dmae"]
    for name, cls in (("crossen", dm.CrossEn), ("negnce", dm.NegNCE)):
        s = W.data_tensor(f"{name}.sim", (7, 7), 0.02).requires_grad_(True)
        loss = cls()(s)
        loss.backward()
        d.update({f"{name}.sim": s, f"{name}.loss": loss, f"{name}.dsim": s.grad})
    save("loss_misc.pt", d)
    L.load_vtp("base_vtp")


# ----------------------------------------------------------------------------- end-to-end fixtures
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


def tiny_clip_batch(bsz, n_clips, seq=12, tag="e2e"):
    img = W.data_tensor(f"{tag}.image", (bsz, n_clips, 3, 32, 32))
    ids = W.data_ints(f"{tag}.ids", (bsz, seq), 1, 300)
    lengths = W.data_ints(f"{tag}.len", (bsz,), 3, seq + 1)
    lengths[0] = seq
    mask = (torch.arange(seq)[None, :] < lengths[:, None]).long()
    ids = ids * mask
    ids[:, 0] = 101
    return dict(
        image=dict(image_data=img, image_pad_mask=torch.zeros(bsz, n_clips, 32, 32, dtype=torch.bool),
                   image_n_clips=[n_clips] * bsz, image_num_frames=[1] * bsz),
        caption=dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids.clone()),
    )


def gen_e2e_clip():
    vtp = L.load_vtp("base_vtp")
    d = {}
    for tag, bsz, n_clips in (("b4n1", 4, 1), ("b3n2", 3, 2)):
        model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(TINY_CLIP_CFG))
        W.fill_module_(model)
        model.train()
        batch = tiny_clip_batch(bsz, n_clips, tag=tag)
        out = model(batch["image"], batch["caption"])
        loss = out["losses"]["level1_similarity_loss"]
        loss.backward()
        cap_input, vis_input, _, _ = model.module.get_l2_input(batch["image"], batch["caption"])
        d.update({
            f"{tag}.image_data": batch["image"]["image_data"],
            f"{tag}.input_ids": batch["caption"]["caption_input_ids"],
            f"{tag}.input_mask": batch["caption"]["caption_input_mask"],
            f"{tag}.loss": loss, f"{tag}.l1_simi": out["l1_simi"],
            f"{tag}.text_embed": cap_input[2], f"{tag}.video_embed": vis_input[2],
        })
        full = ["module.img_encoder.visual.conv1.weight", "module.text_encoder.text_projection",
                "module.img_encoder.visual.proj", "module.img_encoder.visual.class_embedding",
                "module.img_encoder.visual.transformer.resblocks.0.attn.in_proj_bias",
                "module.text_encoder.encoder.layer.1.attention.self.query.bias",
                "module.text_encoder.embeddings.LayerNorm.weight"]
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            d[f"{tag}.gnorm.{n}"] = p.grad.norm()
            d[f"{tag}.gfull.{n}"] = p.grad.to(torch.bfloat16)  # every parameter's gradient (bf16 storage): direction checks, not only norms
            if n in full:
                d[f"{tag}.grad.{n}"] = p.grad
    save("e2e_clip_arch.pt", d)


def gen_temporal_head():
    """UnivlForVideo.get_temporal_output (prj/base_vtp/.../univl_video_pretrain.py:76-90).  The reference hands the [cls] + clip
    features to a HuggingFace AutoModel through `inputs_embeds` (transformers is not pinned: that boundary cannot be executed reproducibly);
    here the method body is executed with the reference's OWN BERT modules in that place -- BertEmbeddings(inputs_embeds, token type 0) followed by
    BertEncoder under the all-ones attention mask, which is what a BERT model does with inputs_embeds -- so the temporal head is pinned by
    reference code up to that substitution."""
    vtp = L.load_vtp("base_vtp")

    class Holder(torch.nn.Module):  # parameter names as in UnivlForVideo: cls_token, temporal_encoder.*
        def __init__(self):
            super().__init__()
            self.cls_token = torch.nn.Parameter(torch.randn(1, 1, 128))
            self.temporal_encoder = vtp["txt_enc"].RobertBertEncoder(
                pretrained=False, vocab_size=40, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, out_dim=128, is_proj=False)
            self.temporal_encoder.embeddings.word_embeddings = None   # add_temporal_head, :71-74
            self.temporal_encoder.module.pooler = None

    m = Holder()
    W.fill_module_(m)
    m.train()
    clip = (W.data_tensor("temporal.clip", (3, 8, 128)) * 0.5).requires_grad_(True)
    w = W.data_tensor("temporal.w", (3, 9, 128))
    bsz, n_clips, _ = clip.shape
    attention_mask = torch.cat((torch.ones(1).expand(bsz, -1), torch.ones((bsz, n_clips), dtype=torch.long)), dim=1)   # :79-83
    input_embeds = torch.cat((m.cls_token.expand(bsz, -1, -1), clip), dim=1)                                              # :85-86
    emb = m.temporal_encoder.embeddings(inputs_embeds=input_embeds, token_type_ids=torch.zeros(bsz, n_clips + 1, dtype=torch.long))
    ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
    seq_out = m.temporal_encoder.encoder(emb, ext, head_mask=[None] * 3)[0]
    (seq_out * w).sum().backward()
    d = {"out": seq_out, "dclip": clip.grad, "dcls": m.cls_token.grad}
    for n, p in m.named_parameters():
        if p.grad is not None:
            d[f"gfull.{n}"] = p.grad.to(torch.bfloat16)
            d[f"gnorm.{n}"] = p.grad.norm()
    save("ops_temporal_head.pt", d)


def gen_vilbert_biattention():
    """antmmf/models/vilbert.py BertBiAttention (:285-416), executed unmodified (only the class body is taken from the file: the
    module's top-level imports pull the whole antmmf package).  Co-attention with 2 heads of 64 and of 128, vision width 96, text width 128,
    ragged key masks on both streams, eval-mode (dropout off) outputs and every gradient."""
    import math as _math
    src = open(f"{L.REF}/antmmf/models/vilbert.py").read()
    body = src[src.index("class BertBiAttention(nn.Module):"):src.index("class BertBiOutput(nn.Module):")]
    ns = {"nn": torch.nn, "torch": torch, "math": _math}
    exec(body, ns)
    d = {}
    # two head sizes: 64 (2 heads over 128) and 128 (2 heads over 256 -- ViLBERT's own bi_hidden_size 1024 / 8 heads ratio); keys of the second "h128."
    for prefix, bi_hidden in (("", 128), ("h128.", 256)):
        cfg = L.AttrDict(dict(bi_hidden_size=bi_hidden, bi_num_attention_heads=2, v_hidden_size=96, hidden_size=128, v_attention_probs_dropout_prob=0.1,
                              attention_probs_dropout_prob=0.1, visualization=False))
        m = ns["BertBiAttention"](cfg)
        W.fill_module_(m)
        m.eval()
        B, Nv, Nt = 3, 9, 14
        x1 = (W.data_tensor("bi.x1", (B, Nv, 96)) * 0.7).requires_grad_(True)
        x2 = (W.data_tensor("bi.x2", (B, Nt, 128)) * 0.7).requires_grad_(True)
        len1, len2 = torch.tensor([9, 4, 7]), torch.tensor([14, 14, 5])
        m1 = ((torch.arange(Nv)[None] >= len1[:, None]).float() * -10000.0)[:, None, None, :]
        m2 = ((torch.arange(Nt)[None] >= len2[:, None]).float() * -10000.0)[:, None, None, :]
        c1, c2, _ = m(x1, m1, x2, m2)
        g1, g2 = W.data_tensor("bi.g1", (B, Nt, bi_hidden)), W.data_tensor("bi.g2", (B, Nv, bi_hidden))
        ((c1 * g1).sum() + (c2 * g2).sum()).backward()
        one = {"x1": x1.detach(), "x2": x2.detach(), "mask1": m1, "mask2": m2, "ctx1": c1, "ctx2": c2, "g1": g1, "g2": g2, "dx1": x1.grad, "dx2": x2.grad}
        one.update(grads_of(m))
        d.update({prefix + k: v for k, v in one.items()})
    save("ops_vilbert_biattention.pt", d)


def gen_m2_eval_recall():
    """prj/M2_Encoder/eval_retrieval.py calu_recall + get_data (:16-127), executed from the reference file (its module-level nn4k imports are
    cut off): random L2-normalised features with 3 captions per image, the printed recalls, and the ground-truth matrices of a small jsonl."""
    import contextlib
    import io
    import json
    import tempfile
    from collections import defaultdict
    import numpy as np

    src = open(f"{L.REF}/prj/M2_Encoder/eval_retrieval.py").read()
    body = src[src.index("def _preprocess_text"):src.index('if __name__ == "__main__":')]
    ns = {"np": np, "torch": torch, "json": json, "defaultdict": defaultdict}
    exec(body, ns)
    n_img, cap = 40, 3
    img = torch.nn.functional.normalize(W.data_tensor("m2eval.img", (n_img, 32)), dim=-1)
    txt = torch.nn.functional.normalize(img.repeat_interleave(cap, 0) * 0.6 + W.data_tensor("m2eval.txt", (n_img * cap, 32)) * 0.25, dim=-1)
    t2i_gt = np.zeros((n_img * cap, n_img)); i2t_gt = np.zeros((n_img, n_img * cap))
    for i in range(n_img):
        for c in range(cap):
            t2i_gt[i * cap + c, i] = 1; i2t_gt[i, i * cap + c] = 1
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns["calu_recall"](txt.numpy(), img.numpy(), t2i_gt, i2t_gt)
    lines = buf.getvalue().strip().splitlines()
    vals = {ln.split()[0]: [float(x) for x in ln.split()[1:]] for ln in lines}
    rows = [{"image": f"img{i}.jpg", "caption": [f"A \u201cCaption\u201d {i} {c}" for c in range(2)]} for i in range(5)]
    with tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False) as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")
    texts, images, t2i, i2t = ns["get_data"](f.name)
    save("metric_m2_recall.pt", {"img": img, "txt": txt, "t2i_gt": torch.from_numpy(t2i_gt).float(), "i2t_gt": torch.from_numpy(i2t_gt).float(),
                                 "t2i_topk": torch.tensor(vals["t2i_topk"]), "i2t_topk": torch.tensor(vals["i2t_topk"]), "MR": torch.tensor(vals["MR"][0]),
                                 "jsonl": [json.dumps(r, ensure_ascii=False) for r in rows], "texts": texts, "images": images,
                                 "data.t2i_gt": torch.from_numpy(t2i).float(), "data.i2t_gt": torch.from_numpy(i2t).float()})


def gen_e2e_clip_stage2():
    vtp = L.load_vtp("base_vtp")
    d = {}
    bsz, n_clips = 4, 2
    batch = tiny_clip_batch(bsz, n_clips, tag="s2")
    d.update({"s2.image_data": batch["image"]["image_data"], "s2.input_ids": batch["caption"]["caption_input_ids"],
              "s2.input_mask": batch["caption"]["caption_input_mask"]})
    for tag, extra in (("plain", {}), ("mine", dict(hard_example_mining=True, re_sample_method="top_k", re_weight_method="median"))):
        cfg = dict(TINY_CLIP_CFG, training_stage="stage1+stage2", with_cross_encoder=True, **extra)
        model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(cfg))
        W.fill_module_(model)
        model.train()
        model.dropout.p = 0.0  # the only stochastic op on the path (BERT dropouts are 0 in this config)
        out = model(batch["image"], batch["caption"])
        loss = out["losses"]["level1_similarity_loss"] + out["losses"]["level2_similarity_loss"]
        if tag == "plain":
            out["l2_simi"].retain_grad()   # (reduce_clips is the identity at level 2: this IS the matrix the level-2 loss reads)
        loss.backward(retain_graph=(tag == "plain"))
        c = out["l2_simi"].grad.detach().clone() if tag == "plain" else None   # (before the pin backward below accumulates into it)
        d.update({f"s2.{tag}.loss1": out["losses"]["level1_similarity_loss"], f"s2.{tag}.loss2": out["losses"]["level2_similarity_loss"],
                  f"s2.{tag}.l2_simi": out["l2_simi"], f"s2.{tag}.l1_simi": out["l1_simi"]})
        for n, p in model.named_parameters():
            if p.grad is not None:
                d[f"s2.{tag}.gnorm.{n}"] = p.grad.norm()
                if tag == "plain":
                    d[f"s2.{tag}.gfull.{n}"] = p.grad.to(torch.bfloat16)
        if tag == "plain":
            # a second scalar of the same graph whose gradient does NOT cancel (the level-2 loss is a softmax over nearly identical pair
            # scores: its parameter gradient is a difference of almost equal terms and bf16 noise dominates its direction): fixed POSITIVE
            # weights on the cross-encoder pair scores
            model.zero_grad(set_to_none=True)
            pin = (out["l2_simi"] * (W.data_tensor("s2.pin", tuple(out["l2_simi"].shape)).abs() + 0.5)).sum()
            pin.backward(retain_graph=True)
            d["s2.pin.value"] = pin.detach()
            for n, p in model.named_parameters():
                if p.grad is not None:
                    d[f"s2.pin.gfull.{n}"] = p.grad.to(torch.bfloat16)
            # the level-2 LOSS gradient, decomposed so that it can be checked tightly: c = d loss / d l2_simi (rows and columns of a softmax
            # gradient: mixed signs, sums ~ 0), loss gradient = J^T c = J^T c+ - J^T c- with c+ = max(c, 0), c- = max(-c, 0).  Each half is a
            # scalar with NON-NEGATIVE weights on the pair scores (no cancellation: checkable at the pin's gates); the subtraction is exact.
            d["s2.plain.dl2_simi"] = c
            for sign, key in ((1.0, "pinp"), (-1.0, "pinm")):
                model.zero_grad(set_to_none=True)
                (out["l2_simi"] * (sign * c).clamp(min=0)).sum().backward(retain_graph=True)
                for n, p in model.named_parameters():
                    if p.grad is not None:
                        d[f"s2.{key}.gfull.{n}"] = p.grad.to(torch.bfloat16)
    save("e2e_clip_stage2.pt", d)


def gen_cnvid_gate():
    """The scheduled hard-mining gate of the CN-VID project (prj/cnvid_vtp/roi_univl/univl/model/univl_video_ret.py:398-428): the reference's own
    forward_stage(..., incre_num) run for a grid of (seed, incre_num); recorded: which similarity routine each call took (mined / plain), the draw
    it made, and the level-2 loss of the plain branch (must equal e2e_clip_stage2.pt's plain loss: same towers, same batch)."""
    vtp = L.load_vtp("cnvid_vtp")
    bsz, n_clips = 4, 2
    batch = tiny_clip_batch(bsz, n_clips, tag="s2")
    cfg = dict(TINY_CLIP_CFG, training_stage="stage1+stage2", with_cross_encoder=True, hard_example_mining=True, re_sample_method="top_k", re_weight_method="median",
               change_iter=5000, change_rate=0.15)
    model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(cfg))
    W.fill_module_(model)
    model.train()
    model.dropout.p = 0.0
    calls = []
    mined, plain = model._cross_similarity_hard_mining, model.get_simi_logits
    model._cross_similarity_hard_mining = lambda *a, **k: (calls.append(1), mined(*a, **k))[1]
    model.get_simi_logits = lambda vis, cap, level, cal: (calls.append(0) if level == "l2" else None, plain(vis, cap, level, cal))[1]
    seeds, incs = list(range(16)), [0.0, 0.15, 0.45, 0.75, 1.0]
    dec = torch.zeros(len(seeds), len(incs), dtype=torch.long)
    draws = torch.zeros(len(seeds))
    loss2_plain = None
    with torch.no_grad():
        cap_input, vis_input, _, _ = model.module.get_l2_input(batch["image"], batch["caption"])
        cap_input, vis_input = cap_input + (batch["caption"],), vis_input + (batch["image"],)
        for si, s in enumerate(seeds):
            torch.manual_seed(s)
            draws[si] = torch.randint(low=0, high=100, size=[1], dtype=torch.float32)[0]
            for ii, inc in enumerate(incs):
                del calls[:]
                torch.manual_seed(s)
                out = model.forward_stage(cap_input, vis_input, True, incre_num=inc)
                assert len(calls) == 1, calls
                dec[si, ii] = calls[0]
                if calls[0] == 0 and loss2_plain is None:
                    loss2_plain = out["losses"]["level2_similarity_loss"].clone()
    assert bool((dec[:, 0] == 0).all()) and 0 < int(dec.sum()) < dec.numel()
    save("e2e_cnvid_gate.pt", {"seeds": torch.tensor(seeds), "incre_num": torch.tensor(incs, dtype=torch.float64), "mined": dec, "draw": draws, "plain.loss2": loss2_plain})


DMAE_E2E = dict(l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="meanP", l3_partial_type=-1, l3_max_frames=4,
                l3_max_words=12, l3_sim_header_hidden_layer=2)


def gen_e2e_dmae_stage3():
    vtp = L.load_vtp("dmae_vtp")
    d = {}
    bsz, n_clips = 4, 4
    batch = tiny_clip_batch(bsz, n_clips, tag="s3")
    d.update({"s3.image_data": batch["image"]["image_data"], "s3.input_ids": batch["caption"]["caption_input_ids"],
              "s3.input_mask": batch["caption"]["caption_input_mask"]})
    for loss_type in ("negNCE", "cross_entropy"):
        cfg = dict(TINY_CLIP_CFG, training_stage="stage1+stage3", l3_loss_type=loss_type, **DMAE_E2E)
        model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(cfg))
        W.fill_module_(model)
        model.train()
        out = model(batch["image"], batch["caption"])
        loss = out["losses"]["level1_similarity_loss"] + out["losses"]["level3_similarity_loss"]
        loss.backward()
        d.update({f"s3.{loss_type}.loss1": out["losses"]["level1_similarity_loss"], f"s3.{loss_type}.loss3": out["losses"]["level3_similarity_loss"],
                  f"s3.{loss_type}.l3_simi": out["l3_simi"]})
        for n, p in model.named_parameters():
            if p.grad is not None:
                d[f"s3.{loss_type}.gnorm.{n}"] = p.grad.norm()
                if loss_type == "negNCE":
                    d[f"s3.{loss_type}.gfull.{n}"] = p.grad.to(torch.bfloat16)
    # the same step with TPM-CL switched on (l3_partial_type 4): losses + the new head's gradient norms only
    cfg = dict(TINY_CLIP_CFG, training_stage="stage1+stage3", l3_loss_type="negNCE", **dict(DMAE_E2E, l3_partial_type=4))
    model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(cfg))
    W.fill_module_(model)
    model.dmae_utils.tis_selector.thresh.fill_(0.6)
    model.train()
    out = model(batch["image"], batch["caption"])
    (out["losses"]["level1_similarity_loss"] + out["losses"]["level3_similarity_loss"]).backward()
    d["s3.tpm4.loss3"] = out["losses"]["level3_similarity_loss"]
    for n, p in model.named_parameters():
        if p.grad is not None and ("xwp" in n or "weight_fc" in n or n.endswith("text_projection") or n.endswith("visual.proj")):
            d[f"s3.tpm4.gnorm.{n}"] = p.grad.norm()
    save("e2e_dmae_stage3.pt", d)


def moco_queue(name, dim, K):
    return torch.nn.functional.normalize(W.data_tensor(name, (dim, K)), dim=0)


def perturb_towers_(module, scale=1.05):
    """Stands in for an optimizer step between the two MoCo steps (so that the momentum update has something to average)."""
    with torch.no_grad():
        for p in module.parameters():
            p.mul_(scale)


def gen_e2e_clip_moco():
    vtp = L.load_vtp("base_vtp")
    cfg = dict(TINY_CLIP_CFG, with_moco=True, K=64, M=0.5)
    d = {}
    bsz, n_clips, tag = 4, 2, "moco"
    model = vtp["ret"].UnivlForVideoTextRetrieval(L.AttrDict(cfg))
    W.fill_module_(model)
    model.train()
    mu = vtp["moco"].MocoUtils(L.AttrDict(cfg), img_encoder=model.module.img_encoder, txt_encoder=model.module.text_encoder)
    mu.txt_queue.copy_(moco_queue("moco.txt_queue", 128, 64))
    mu.img_queue.copy_(moco_queue("moco.img_queue", 128, 16384))
    model.moco_utils = mu
    batch = tiny_clip_batch(bsz, n_clips, tag=tag)
    d.update({f"{tag}.image_data": batch["image"]["image_data"], f"{tag}.input_ids": batch["caption"]["caption_input_ids"],
              f"{tag}.input_mask": batch["caption"]["caption_input_mask"]})
    for step in (1, 2):
        model.zero_grad(set_to_none=True)
        out = model(batch["image"], batch["caption"])
        loss = out["losses"]["level1_similarity_loss"]
        loss.backward()
        d[f"{tag}.loss{step}"] = loss
        d[f"{tag}.l1_simi{step}"] = out["l1_simi"]
        for n, p in model.named_parameters():
            if p.grad is not None:
                d[f"{tag}.gnorm{step}.{n}"] = p.grad.norm()
        d[f"{tag}.txt_queue_head{step}"] = mu.txt_queue[:, :12].clone()
        d[f"{tag}.img_queue_head{step}"] = mu.img_queue[:, :20].clone()
        d[f"{tag}.txt_ptr{step}"] = mu.txt_queue_ptr.clone()
        d[f"{tag}.img_ptr{step}"] = mu.img_queue_ptr.clone()
        if step == 1:
            perturb_towers_(model.module)
    d[f"{tag}.key_probe"] = dict(mu.txt_encoder_k.named_parameters())["encoder.layer.0.attention.self.query.weight"][:4, :8].clone()
    save("e2e_clip_moco.pt", d)


TINY_M2 = dict(beit_version="base", encoder_embed_dim=128, out_embed_dim=64, encoder_layers=2, beit3_vl_layers=1,
               image_size=32, patch_size=8, vocab_size=300, max_text_len=12, encoder_attention_heads=2)


def m2_config():
    m2 = L.load_m2()
    src = m2["cfg_src"]
    body = src[src.index("def config():"):]
    end = body.index("\n@ex.named_config") if "\n@ex.named_config" in body else len(body)
    ns = {"_loss_names": m2["cfgmod"]._loss_names}
    exec(body[:end].replace("def config():", "def config():\n    pass", 1) + "\n    return dict(locals())\n", ns)
    cfg = ns["config"]()
    cfg.update(TINY_M2)
    cfg.update(loss_names=m2["cfgmod"]._loss_names({"itc": 1}), test_only=True, load_path="",
               tokenizer=m2["root"] + "/vlmo/tokenizer", tokenizer_type="GLMChineseTokenizer")
    return cfg


def gen_e2e_m2():
    m2 = L.load_m2()
    cfg = m2_config()
    model = m2["VLMo"](cfg)
    W.fill_module_(model)
    model.eval()  # dropout is 0 everywhere in the M2 configs; eval == train numerically
    img = (W.data_tensor("m2.image", (3, 3, 32, 32)) * 0.25 + 0.5).clamp(0, 1)
    ids = W.data_ints("m2.ids", (3, 12), 1, 300)
    lengths = torch.tensor([12, 5, 8])
    mask = (torch.arange(12)[None, :] < lengths[:, None]).long()
    ids = ids * mask
    img.requires_grad_(False)
    out_i = model.infer_image({"image": [img]})
    out_t = model.infer_text({"text_ids": ids, "text_labels": ids, "text_masks": mask})
    ls, lvs = model.logit_scale.exp(), model.logit_vl_scale.exp()
    logits = ls * out_i["cls_feats"] @ out_t["cls_feats"].t()
    logits_vl = lvs * out_i["cls_vlffn_feats"] @ out_t["cls_vlffn_feats"].t()
    # scalar used only to pin gradients of the towers (the reference ships no M2 training loss)
    pin = (logits * W.data_tensor("m2.wl", (3, 3))).sum() + (logits_vl * W.data_tensor("m2.wvl", (3, 3))).sum()
    pin.backward()
    d = {"image": img, "text_ids": ids, "text_masks": mask,
         "img.cls_feats": out_i["cls_feats"], "img.cls_vlffn_feats": out_i["cls_vlffn_feats"],
         "img.image_feats": out_i["image_feats"],
         "txt.cls_feats": out_t["cls_feats"], "txt.cls_vlffn_feats": out_t["cls_vlffn_feats"],
         "logits": logits, "logits_vl": logits_vl, "pin": pin}
    full = ["backbone.vision_embed.proj.weight", "itc_image_proj.fc.weight", "itc_vl_text_proj.fc.weight",
            "backbone.encoder.layers.0.self_attn.q_proj.B.bias", "backbone_vl.layers.0.ffn.A.ffn_layernorm.weight",
            "logit_scale"]
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        d[f"gnorm.{n}"] = p.grad.norm()
        d[f"gfull.{n}"] = p.grad.to(torch.bfloat16)
        if n in full:
            d[f"grad.{n}"] = p.grad
    d["param_names"] = [n for n, _ in model.named_parameters()]
    save("e2e_m2.pt", d)


# ----------------------------------------------------------------------------- distributed fixture
def _gather_worker(rank, world, port, ret):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    du = L.load_antmmf_core()["du"]
    full_t = W.data_tensor("gather.t", (world * 3, 8))
    full_v = W.data_tensor("gather.v", (world * 3, 8))
    t = full_t[rank * 3:(rank + 1) * 3].clone().requires_grad_(True)
    v = full_v[rank * 3:(rank + 1) * 3].clone().requires_grad_(True)
    gt = du.gather_tensor(t, method="cat", back_gradient=True, pad_tensors=True)
    gv = du.gather_tensor(v, method="cat", back_gradient=True, pad_tensors=True)
    Ret = L.load_vtp("base_vtp")["ret"].UnivlForVideoTextRetrieval
    loss = Ret.get_mil_nce_loss(None, gt @ gv.t(), world * 3, 1)
    loss.backward()
    ret[rank] = dict(loss=loss.detach(), dt=t.grad.clone(), dv=v.grad.clone())
    dist.destroy_process_group()


def gen_gather():
    import torch.multiprocessing as mp

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(world, 29611, ret), nprocs=world, join=True)
    d = {"world": world}
    for r in range(world):
        for k, v in ret[r].items():
            d[f"rank{r}.{k}"] = v
    save("gather_w2.pt", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["clip_block", "bert_layer", "m2_layer", "dmae_seqtransf", "dmae_wti", "dmae_tpmcl", "metric_recall", "m2_eval_recall", "m2_ckpt_convert", "losses", "e2e_clip", "temporal_head", "vilbert_biattention", "e2e_clip_stage2", "cnvid_gate", "e2e_dmae_stage3", "e2e_clip_moco", "e2e_m2", "gather"]
    fns = dict(m2_eval_recall=gen_m2_eval_recall, vilbert_biattention=gen_vilbert_biattention, temporal_head=gen_temporal_head, dmae_tpmcl=gen_dmae_tpmcl, m2_ckpt_convert=gen_m2_ckpt_convert, metric_recall=gen_metric_recall, dmae_seqtransf=gen_dmae_seqtransf, dmae_wti=gen_dmae_wti, clip_block=gen_clip_block, bert_layer=gen_bert_layer, m2_layer=gen_m2_layer, losses=gen_losses,
               e2e_clip=gen_e2e_clip, e2e_clip_moco=gen_e2e_clip_moco, e2e_clip_stage2=gen_e2e_clip_stage2, cnvid_gate=gen_cnvid_gate, e2e_dmae_stage3=gen_e2e_dmae_stage3, e2e_m2=gen_e2e_m2, gather=gen_gather)
    for w in which:
        fns[w]()
