"""Image-text retrieval evaluation of the M2 encoder on the MI355X path (reference: prj/M2_Encoder/eval_retrieval.py:10-127).

Same functions and meaning -- `get_data` (jsonl of {"image", "caption": [...]} -> texts, images and the two 0/1 ground-truth matrices),
`extract_feats`, `calu_recall` (text->image and image->text recall@1/5/10 in percent and their mean "MR") -- re-designed for the device:
the reference pushes ONE sample at a time through the nn4k invoker, moves every feature to numpy, argsorts both full similarity matrices on
the CPU and walks the top-10 in Python; here captions / images go through `VLMo.infer_text / infer_image` in batches, the similarity matrix
stays on the GPU and "is a ground truth among the top k" is the rank of the best ground truth from the rank-counting kernel
(`antmmf_rank_rows`, ties by index like a stable argsort)."""
import json
from collections import defaultdict

import torch

from antmmf.hip import contrastive
from antmmf.modules.metrics.global_retrieval_recall import gt_ranks


def _preprocess_text(text):
    return text.lower().replace("“", '"').replace("”", '"')  # adapt the text to the Chinese BERT vocab (reference :10-13)


def get_data(data_file):
    """-> texts, images, txt2img_gt [T, I], img2txt_gt [I, T]  (reference :16-46)."""
    img2txt, txt2img = defaultdict(list), defaultdict(list)
    texts, images, text_ids, image_ids = [], [], {}, {}
    with open(data_file, "r") as f:
        for i, line in enumerate(f):
            data = json.loads(line.strip())
            img, cap = data["image"], data["caption"]
            images.append(img)
            image_ids[img] = i
            for c in cap:
                c = _preprocess_text(c)
                img2txt[img].append(c)
                txt2img[c].append(img)
                texts.append(c)
                text_ids[c] = len(texts) - 1
    img2txt_gt = torch.zeros(len(images), len(texts))
    txt2img_gt = torch.zeros(len(texts), len(images))
    for i, img in enumerate(images):
        for txt in img2txt[img]:
            img2txt_gt[i, text_ids[txt]] = 1
    for i, txt in enumerate(texts):
        for img in txt2img[txt]:
            txt2img_gt[i, image_ids[img]] = 1
    return texts, images, txt2img_gt, img2txt_gt


@torch.no_grad()
def extract_feats(model, text_batches, image_batches, feat_key="cls_vlffn_feats"):
    """text_batches: iterable of dicts {"text_ids", "text_masks"}; image_batches: iterable of image tensors [b, 3, H, W] in [0, 1].
    Returns (txt_feats [T, D], img_feats [I, D]) on the device (`itc_feats_name` of the reference config: cls_vlffn_feats)."""
    was_training = model.training
    model.eval()
    txt = [model.infer_text({"text_ids": b["text_ids"], "text_masks": b["text_masks"]})[feat_key].float() for b in text_batches]
    img = [model.infer_image({"image": [b]})[feat_key].float() for b in image_batches]
    if was_training:
        model.train()
    return torch.cat(txt, 0), torch.cat(img, 0)


def _gt_lists(gt):
    return [torch.nonzero(row, as_tuple=False).flatten().tolist() for row in gt]


def calu_recall(txt_feats, img_feats, txt2img_gt, img2txt_gt, verbose=True):
    """Recall@{1, 5, 10} in percent for both directions and their mean (reference :70-127: a hit at k = some ground truth among the k best)."""
    t2i = contrastive.matmul_f32(txt_feats.float(), img_feats.float())   # fp32-accurate similarities on the MFMA GEMM (the library refuses host tensors)
    rt = gt_ranks(t2i, _gt_lists(txt2img_gt)).float()
    ri = gt_ranks(t2i.t().contiguous(), _gt_lists(img2txt_gt)).float()
    t2i_topk = [float((rt < k).float().mean()) * 100 for k in (1, 5, 10)]
    i2t_topk = [float((ri < k).float().mean()) * 100 for k in (1, 5, 10)]
    mr = sum(t2i_topk + i2t_topk) / 6
    if verbose:
        print("t2i_topk", *[round(x, 1) for x in t2i_topk])
        print("i2t_topk", *[round(x, 1) for x in i2t_topk])
        print("MR", round(mr, 1))
    return {"t2i_r@1": t2i_topk[0], "t2i_r@5": t2i_topk[1], "t2i_r@10": t2i_topk[2], "i2t_r@1": i2t_topk[0], "i2t_r@5": i2t_topk[1],
            "i2t_r@10": i2t_topk[2], "MR": mr}
