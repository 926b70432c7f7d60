"""oracle.step -- CPU restatement of the whole contrastive step (towers -> embeddings -> gather ->
similarity -> loss).  TEST INFRASTRUCTURE ONLY (see oracle/ops.py)."""
import torch

from . import losses, ops, towers


def univl_stage1(P, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads,
                 gather=None, training=True):
    """UnivlForVideoTextRetrieval stage-1 step, arch_type "clip", no MoCo
    (prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:357-387,457-480;
    univl_video_base.py:56-166,272-299).

    image_data [B, n_clips*f, 3, H, W] with f = 1 frame per clip; P holds the reference's state_dict
    (keys prefixed "module.").  `gather(t)` maps a local [b, D] tensor to the global [B_g, D]
    (identity when None).  Returns dict(loss, l1_simi [T, V], text_embed, video_embed)."""
    b, t = image_data.shape[:2]
    frames_per_clip = t // n_clips
    vis = towers.clip_vit(towers._sub(P, "module.img_encoder.visual."), image_data.flatten(0, 1), vit_heads, patch)
    d = vis.shape[-1]
    # mean over the frames of a clip (the grid is 1x1 and unmasked for the ViT encoder: clip_visual_encoder.py:86-94,
    # univl_video_base.py:78-97), then L2 normalise each clip feature (:114)
    clip_feat = vis.view(b * n_clips, frames_per_clip, d).mean(1)
    video_embed = ops.l2_normalize(clip_feat)
    _, pooled = towers.roberta_bert_encoder(towers._sub(P, "module.text_encoder."), input_ids, input_mask, bert_heads)
    text_embed = ops.l2_normalize(pooled)
    g_text = gather(text_embed) if gather else text_embed
    g_video = gather(video_embed) if gather else video_embed
    bg = g_text.shape[0]
    # [V, n, D] x [D, T] -> [T, V, n]   (univl_video_ret.py:208-213)
    simi = torch.matmul(g_video.view(bg, n_clips, d), g_text.t()).permute(2, 0, 1)
    out = dict(text_embed=text_embed, video_embed=video_embed, l1_simi=torch.logsumexp(simi, dim=-1))
    if training:
        # tile every text row n times: [T*n, V*n]  (univl_video_ret.py:375-379)
        mil = simi.unsqueeze(1).expand(bg, n_clips, bg, n_clips).reshape(bg * n_clips, bg * n_clips)
        out["loss"] = losses.mil_nce(mil, bg, n_clips)
    return out


def univl_stage1_moco(P, Pk, queues, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads,
                      momentum=0.9999, temperature=0.05):
    """One training step of the level-1 loss with MoCo (univl_video_ret.py:262-312; moco_utils.py:55-107), single process.
    P: query parameters (autograd leaves), Pk: key-tower parameters (same names; updated IN PLACE by the momentum rule),
    queues: dict(txt=[D, K_t], img=[D, K_i], txt_ptr=int, img_ptr=int), updated in place by dequeue_and_enqueue.
    Order as in the reference: momentum update -> key features (no grad) -> both losses -> enqueue."""
    with torch.no_grad():
        for k in Pk:
            Pk[k] = Pk[k] * momentum + P[k].detach() * (1.0 - momentum)
    q = univl_stage1(P, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads, training=False)
    with torch.no_grad():
        kk = univl_stage1(Pk, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads, training=False)
    q_v, q_t, key_v, key_t = q["video_embed"], q["text_embed"], kk["video_embed"], kk["text_embed"]
    pos = (q_v * key_t.repeat_interleave(n_clips, 0)).sum(-1, keepdim=True)          # [B*n, 1]
    loss_v = losses.moco(pos, q_v @ queues["txt"].clone(), temperature)
    pos = (q_t.repeat_interleave(n_clips, 0) * key_v).sum(-1).view(-1, n_clips)      # [B, n]
    loss_t = losses.moco(pos, q_t @ queues["img"].clone(), temperature)
    for keys, nm in ((key_v, "img"), (key_t, "txt")):
        K, ptr, n = queues[nm].shape[1], queues[nm + "_ptr"], keys.shape[0]
        end = min(ptr + n, K)
        queues[nm][:, end - n:end] = keys.t()
        queues[nm + "_ptr"] = end % K
    return dict(loss=(loss_t + loss_v) / 2.0, l1_simi=q["l1_simi"], text_embed=q_t, video_embed=q_v)


def univl_stage2(P, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads, chosen=None, weight=None):
    """Stage-2 cross-encoder scoring + MIL-NCE on the [T, V] score matrix (univl_video_ret.py:33-144,389-443;
    univl_video_base.py:168-271), single process, dropout p = 0.
    Every (caption t, video v) pair: [BertEmbeddings(ids_t, type 0) ; BertEmbeddings(inputs_embeds=[clip tokens_v ; word_emb[102]],
    type 1)] -> the text tower's BERT layers with the -10000 key mask -> cls @ text_projection -> similarity_dense.
    `chosen` [T, T] (optional): hard-negative mining's video index for every (t, column) -- row t is scored against
    videos chosen[t, :] (the reference picks them with torch.topk(sorted=False), whose order is unspecified; the caller passes
    the indices actually used).  `weight` [T]: optional per-row loss weights (:192-195).  Returns dict(l2_simi, loss)."""
    b = image_data.shape[0]
    frames_per_clip = image_data.shape[1] // n_clips
    Pt = towers._sub(P, "module.text_encoder.")
    Pe = towers._sub(Pt, "embeddings.")
    vis = towers.clip_vit(towers._sub(P, "module.img_encoder.visual."), image_data.flatten(0, 1), vit_heads, patch)
    clip_tokens = vis.view(b * n_clips, frames_per_clip, -1).mean(1).view(b, n_clips, -1)          # visual_embed (not normalised)
    cap_embed = towers.bert_embeddings(Pe, input_ids=input_ids)                                   # [T, S, h]
    sep = Pe["word_embeddings.weight"][torch.full((b,), 102)].unsqueeze(1)
    vis_in = torch.cat([clip_tokens, sep], 1)
    vis_embed = towers.bert_embeddings(Pe, inputs_embeds=vis_in, token_type_ids=torch.ones(vis_in.shape[:2], dtype=torch.long))
    vis_mask = torch.ones(b, n_clips + 1)
    if chosen is None:
        chosen = torch.arange(b)[None, :].expand(b, b)
    rows = []
    for t in range(b):
        idx = chosen[t]
        emb = torch.cat([cap_embed[t][None].expand(b, -1, -1), vis_embed[idx]], 1)
        mask = torch.cat([input_mask[t][None].expand(b, -1).float(), vis_mask[idx]], 1)
        seq = towers.bert_encoder(towers._sub(Pt, "encoder."), emb, towers.bert_key_bias(mask), bert_heads)
        pooled = seq[:, 0] @ Pt["text_projection"]
        hid = torch.relu(ops.linear(pooled, P["similarity_dense.0.weight"], P["similarity_dense.0.bias"]))
        rows.append(ops.linear(hid, P["similarity_dense.2.weight"], P["similarity_dense.2.bias"]).view(1, b))
    l2 = torch.cat(rows, 0)
    return dict(l2_simi=l2, loss=losses.mil_nce(l2, b, 1, weight))


def dmae_stage3(P, image_data, input_ids, input_mask, n_clips, vit_heads, patch, bert_heads, loss_type="negNCE", interaction="wti",
                with_va=True, sim_header="meanP", sim_layers=2):
    """DMAE stage-3 head on the clip-arch towers (prj/dmae_vtp/roi_univl/univl/model/univl_video_ret.py:457-476;
    dmae_utils.py:249-278,229-247,133-184), single process, l3_partial_type -1.
    Word features = BertEmbeddings(ids) (the EMBEDDING layer output, univl_video_base.py:168-176), frame features =
    BertEmbeddings(inputs_embeds=[clip tokens ; word_emb[102]], type 1) (:178-204), both L2-normalised per token; sentence feature
    = the stage-1 pooled text embedding; optional seqTransf aggregation; token-wise interaction -> [T, V]; loss in both directions."""
    b = image_data.shape[0]
    frames_per_clip = image_data.shape[1] // n_clips
    Pt = towers._sub(P, "module.text_encoder.")
    Pe = towers._sub(Pt, "embeddings.")
    vis = towers.clip_vit(towers._sub(P, "module.img_encoder.visual."), image_data.flatten(0, 1), vit_heads, patch)
    clip_tokens = vis.view(b * n_clips, frames_per_clip, -1).mean(1).view(b, n_clips, -1)
    cap_embed = towers.bert_embeddings(Pe, input_ids=input_ids)
    sep = Pe["word_embeddings.weight"][torch.full((b,), 102)].unsqueeze(1)
    vis_in = torch.cat([clip_tokens, sep], 1)
    vis_embed = towers.bert_embeddings(Pe, inputs_embeds=vis_in, token_type_ids=torch.ones(vis_in.shape[:2], dtype=torch.long))
    vis_mask = torch.ones(b, n_clips + 1)
    _, pooled = towers.roberta_bert_encoder(Pt, input_ids, input_mask, bert_heads)
    return dmae_stage3_head(P, cap_embed, vis_embed, vis_mask, ops.l2_normalize(pooled), input_mask, loss_type, interaction, with_va, sim_header, sim_layers)


def dmae_stage3_head(P, cap_embed, vis_embed, vis_mask, text_l1, input_mask, loss_type="negNCE", interaction="wti", with_va=True, sim_header="meanP", sim_layers=2):
    """The stage-3 head alone, from the token features up (dmae_utils.py:249-278: per-token L2 normalisation, optional seqTransf aggregation, token-wise interaction,
    loss in both directions): cap_embed [B, T, D] word features, vis_embed [B, V, D] frame features, text_l1 [B, D] the L2-normalised stage-1 sentence embedding.
    Split out of dmae_stage3 so that the tests can evaluate the fp32 head on the PRODUCT's tower outputs (which part of a level-3 deviation is the towers' bf16 and which
    the head's)."""
    text_l1 = text_l1.unsqueeze(1)
    cap_embed = cap_embed / cap_embed.norm(dim=-1, keepdim=True)
    vis_embed = vis_embed / vis_embed.norm(dim=-1, keepdim=True)
    Pd = towers._sub(P, "dmae_utils.")
    agg, tok_mask, _ = towers.dmae_agg_visual_feat(Pd, vis_embed, vis_mask, heads=vis_embed.shape[-1] // 64, layers=sim_layers, sim_header=sim_header)
    simi = losses.dmae_wti_interaction(Pd, text_l1, cap_embed, agg, input_mask.float(), tok_mask, interaction, with_va)
    fn = losses.neg_nce if loss_type == "negNCE" else losses.cross_en
    # feats: what DmaeUtils.get_partial_similarity (TPM-CL) is handed next to the score matrix -- the sentence feature [B, 1, D], the normalised word features and the
    # normalised, NOT aggregated frame features (dmae_utils.py:262-277)
    return dict(l3_simi=simi, loss=(fn(simi) + fn(simi.t())) / 2, feats=(text_l1, cap_embed, vis_embed, vis_mask))


def m2_itc(P, image, text_ids, text_masks, heads, patch, gather=None):
    """M2 two-level ITC step: towers pinned by VLMo.infer_image/infer_text, logits formula from
    prj/M2_Encoder/m2_encoder.py:92-95, symmetric CE on both the cls and the cls_vlffn pairs
    (loss form is this build's: see oracle.losses.clip_itc)."""
    oi = towers.m2_infer_image(P, image, heads, patch)
    ot = towers.m2_infer_text(P, text_ids, text_masks, heads)
    g = gather if gather else (lambda x: x)
    l1, logits = losses.clip_itc(g(oi["cls_feats"]), g(ot["cls_feats"]), P["logit_scale"])
    l2, logits_vl = losses.clip_itc(g(oi["cls_vlffn_feats"]), g(ot["cls_vlffn_feats"]), P["logit_vl_scale"])
    return dict(loss=0.5 * (l1 + l2), logits=logits, logits_vl=logits_vl, img=oi, txt=ot)


class GatherWithGrad(torch.autograd.Function):
    """Single-process model of GradientAllGather + gather_tensor(method="cat")
    (antmmf/utils/distributed_utils.py:92-189) for W simulated ranks: forward concatenates the W
    local shards; backward hands rank r the SUM over all ranks' gradients for its shard.  When every
    rank computes the same global loss this is W x the single-process gradient."""

    @staticmethod
    def forward(ctx, world, *shards):
        ctx.world = world
        return torch.cat(shards, dim=0)

    @staticmethod
    def backward(ctx, grad):
        return (None,) + tuple(ctx.world * g for g in grad.chunk(ctx.world, dim=0))
