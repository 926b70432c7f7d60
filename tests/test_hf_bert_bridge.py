"""PretrainedTransformerEncoder (the `arch_type: univl` text tower): a HuggingFace BERT checkpoint loaded key for key into the in-repo fused BERT, against
transformers' own BertModel on the same weights (reference antmmf/modules/encoders/text_encoder.py:32-175).  CPU lane emulator here, MI355X under -m gpu."""
import json
import os
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "build", "libantmmf_emu.so")


def _hf_checkpoint(tmp, layers=3):
    tr = pytest.importorskip("transformers")
    cfg = tr.BertConfig(vocab_size=120, hidden_size=128, num_hidden_layers=layers, num_attention_heads=2, intermediate_size=256, max_position_embeddings=40,
                        type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg._attn_implementation = "eager"   # the sdpa path of recent transformers cannot return attention maps
    torch.manual_seed(7)
    hf = tr.BertModel(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():   # bf16-representable weights: the comparison then sees kernel error only
            p.copy_((p * 3).to(torch.bfloat16).float())
    d = os.path.join(tmp, "tiny-bert")
    os.makedirs(d, exist_ok=True)
    torch.save({"bert." + k: v for k, v in hf.state_dict().items()}, os.path.join(d, "pytorch_model.bin"))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg.to_dict(), f)
    return hf, d


def _case(dev, tmp):
    from antmmf.modules.encoders import TextEncoder

    hf, d = _hf_checkpoint(tmp)
    enc = TextEncoder({"type": "PretrainedTransformerEncoder", "params": dict(pretrained=True, bert_model_name=d, num_hidden_layers=2, start_hidden_layer=1,
                                                                             num_segments=3, hidden_size=128, vocab_size=120)}).module.to(dev).eval()
    assert len(enc.encoder.layer) == 2 and enc.embeddings.token_type_embeddings.num_embeddings == 3 and enc.pooler is enc.module.pooler
    assert torch.equal(enc.embeddings.token_type_embeddings.weight[:2].cpu(), hf.embeddings.token_type_embeddings.weight)
    hf.encoder.layer = torch.nn.ModuleList(list(hf.encoder.layer[1:3]))
    ids = torch.randint(1, 120, (3, 12), generator=torch.Generator().manual_seed(1))
    mask = (torch.arange(12)[None, :] < torch.tensor([12, 7, 4])[:, None]).long()
    ids = ids * mask
    tt = torch.zeros_like(ids)
    with torch.no_grad():
        ref_seq, ref_pool = hf(input_ids=ids, attention_mask=mask, token_type_ids=tt, return_dict=False)[:2]
        seq, pool = enc(input_ids=ids.to(dev), attention_mask=mask.to(dev), token_type_ids=tt.to(dev))
        # output_attentions=True: the fused tower returns the reduction the reference applies to the maps (univl_video_base.py:138-143) -- checked here against that
        # very expression on transformers' own attention maps
        ref_att = hf(input_ids=ids, attention_mask=mask, token_type_ids=tt, return_dict=False, output_attentions=True)[2]
        want_imp = torch.cat([a.mean(1, keepdim=True) for a in ref_att], dim=1).sum(dim=(1, 2))
        out3 = enc(input_ids=ids.to(dev), attention_mask=mask.to(dev), token_type_ids=tt.to(dev), output_attentions=True)
    assert len(out3) == 3 and torch.allclose(out3[0].float().cpu(), seq.float().cpu())
    got_imp = out3[2].value.cpu()
    err_imp = float((got_imp - want_imp).abs().max() / want_imp.abs().max())
    assert got_imp.shape == want_imp.shape and err_imp < 3e-2, err_imp
    keep = mask.bool()[..., None]
    err_seq = float(((seq.float().cpu() - ref_seq) * keep).abs().max() / ref_seq.abs().max())
    err_pool = float((pool.float().cpu() - ref_pool).abs().max())
    assert err_seq < 3e-2 and err_pool < 3e-2, (err_seq, err_pool)
    with pytest.raises(FileNotFoundError):
        TextEncoder({"type": "PretrainedTransformerEncoder", "params": dict(pretrained=True, bert_model_name=os.path.join(tmp, "absent"))})
    return err_seq, err_pool


def _case_univl_arch(dev, stage="stage1+stage2"):
    """`arch_type: univl` through the retrieval model: PretrainedTransformerEncoder text / cross tower with its pooler heads + img_fc; loss finite and at
    ln(B)-scale at random init, every trainable parameter of the new pieces receives a gradient."""
    import copy
    import math

    import model_cases as mc
    import weightgen as W
    from antmmf.common.configuration import Configuration
    from roi_univl.univl.model.univl_video_ret import UnivlForVideoTextRetrieval

    cfg = copy.deepcopy(mc.TINY_CLIP_CFG)
    cfg.update(arch_type="univl", training_stage=stage, with_cross_encoder="stage2" in stage)
    cfg["text_encoder"] = dict(type="PretrainedTransformerEncoder", params=dict(pretrained=False, vocab_size=300, hidden_size=128, intermediate_size=512,
                                                                               num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=40, num_segments=2))
    model = UnivlForVideoTextRetrieval(Configuration(cfg))
    W.fill_module_(model)
    model = model.to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B = 4
    img = (W.data_tensor("univl.arch.image", (B, 1, 3, 32, 32)) * 0.25 + 0.5).clamp(0, 1).to(dev)
    ids = W.data_ints("univl.arch.ids", (B, 12), 1, 300)
    mask = (torch.arange(12)[None, :] < torch.tensor([12, 5, 8, 3])[:, None]).long()
    ids = (ids * mask).to(dev)
    mask = mask.to(dev)
    img_input = dict(image_data=img, image_pad_mask=torch.zeros(B, 1, 32, 32, dtype=torch.bool, device=dev), image_n_clips=[1] * B, image_num_frames=[1] * B)
    cap_input = dict(caption_input_ids=ids, caption_input_mask=mask, caption_raw_input_ids=ids)
    out = model(img_input, cap_input)
    # training + arch univl: the text tower's key importance reaches the model as `words_importance` (reference univl_video_base.py:131-143).  Every softmax row
    # sums to one, so each caption's importances sum to layers x tokens -- up to the attention-probability dropout of the BERT layers (p = 0.1 in training, and
    # the maps HF returns are the dropped ones): 24 +- a few per cent
    assert model.module.forward_text_encoder(ids, mask)["words_importance"] is None     # nobody asked: no extra score pass
    model.module.want_words_importance = True
    wi = model.module.forward_text_encoder(ids, mask)["words_importance"]
    assert wi is not None and tuple(wi.shape) == (B, 12) and not wi.requires_grad
    torch.testing.assert_close(wi.sum(-1).cpu(), torch.full((B,), 2.0 * 12), rtol=0.15, atol=0.0)
    assert float(wi[1, 5:].abs().max()) < 1e-3      # padded words (mask -10000) receive no attention
    loss = sum(out["losses"].values())
    assert torch.isfinite(loss) and 0.2 * math.log(B) < float(out["losses"]["level1_similarity_loss"]) < 6 * math.log(B), out["losses"]
    loss.backward()
    named = dict(model.named_parameters())
    need = [n for n in named if "img_fc" in n or "text_encoder.pooler" in n or ("cross_pooler" in n and "stage2" in stage)]
    assert len(need) >= 6, need
    for n in need:
        assert named[n].grad is not None and float(named[n].grad.abs().sum()) > 0, n
    return {k: float(v) for k, v in out["losses"].items()}


@pytest.fixture()
def emu():
    from test_kernels_emu import _stale

    if _stale():
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    from antmmf.hip import _lib

    old = os.environ.get("ANTMMF_HIP_LIB")
    os.environ["ANTMMF_HIP_LIB"] = EMU_LIB
    _lib.reset_for_tests()
    yield torch.device("cpu")
    if old is None:
        os.environ.pop("ANTMMF_HIP_LIB", None)
    else:
        os.environ["ANTMMF_HIP_LIB"] = old
    _lib.reset_for_tests()


def test_hf_bert_checkpoint_on_fused_bert_emulated(emu, tmp_path):
    print(_case(emu, str(tmp_path)))


@pytest.mark.skipif(not os.environ.get("ANTMMF_SLOW_TESTS"), reason="set ANTMMF_SLOW_TESTS=1 (an emulated model step: ~1.5 min; the same case runs under -m gpu)")
def test_univl_arch_model_emulated(emu):
    print(_case_univl_arch(emu, "stage1"))


@pytest.mark.gpu
def test_hf_bert_checkpoint_on_fused_bert_gpu(tmp_path):
    print(_case(torch.device("cuda:0"), str(tmp_path)))


@pytest.mark.gpu
def test_univl_arch_model_gpu():
    print(_case_univl_arch(torch.device("cuda:0"), "stage1+stage2"))
