"""Default VLMo config keys read on the ITC path (subset of the reference's prj/M2_Encoder/vlmo/config.py:22-165,
same names and defaults) and a loader for the shipped json configs (prj/M2_Encoder/configs/*.json)."""
import json


def _loss_names(d):
    ret = {"itm": 0, "itc": 0, "caption": 0, "mvlm": 0, "textmlm": 0, "imagemlm": 0, "vqa": 0, "nlvr2": 0, "irtr": 0}
    ret.update(d)
    return ret


def default_config():
    return dict(
        exp_name="vlmo", seed=1, loss_names=_loss_names({"itc": 1}), batch_size=1024,
        beit_version="base", encoder_layers=9, encoder_embed_dim=768, out_embed_dim=768, beit3_vl_layers=3,
        image_size=224, patch_size=16, vocab_size=64010, max_text_len=52, drop_path_rate=0.0,
        checkpoint_activations=False, share_layer=False, share_attn=False, deepnorm=False, mask_ratio=0, one_attn=False,
        atorch_config=None, test_only=False, load_path="", tokenizer=None, tokenizer_type=None, cap_onlytext=False, lang="cn",
        num_frames=1, coalesce_backbone=False, mask_data="v+l", itc_mask=False, local_loss=False, aggregate_nodes=-1,
        use_dual_softmax=False, split_data_for_imagemlm=False, log_metric_steps=50, itc_feats_name="cls_vlffn_feats",
    )


def load_json_config(path, **overrides):
    cfg = default_config()
    with open(path, "r", encoding="utf-8") as f:
        cfg.update(json.load(f))
    if isinstance(cfg.get("loss_names"), dict):
        cfg["loss_names"] = _loss_names(cfg["loss_names"])
    cfg.update(overrides)
    return cfg
