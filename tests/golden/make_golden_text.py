"""Golden vectors for the caption processors.  Build container only:  python tests/golden/make_golden_text.py

Executes the reference's own `MaskedTokenProcessor` / `MaskedBertTokenizer` class bodies (read from
/root/reference/antmmf/datasets/processors/text_processors.py at generation time, never stored) on top of transformers' BertTokenizer
built from the reference's test vocabularies -- what `build_tokenizer` hands them for `model_type: bert`
(antmmf/datasets/build.py:91-110) -- and writes inputs + expected tensors to tests/golden/text_processors.json.  The two vocabulary
files under tests/golden/vocabs/ are the reference's test data (tests/data/vocabs/), copied as fixtures.
"""
import json
import os
import random
import sys
import types

import torch

REF = os.environ.get("ANTMMF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
VOCABS = {"uncased": "bert-base-uncased_30522_vocab.txt", "chinese": "bert-base-chinese_21128_vocab.txt"}


class Cfg(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) else v

    def get(self, k, d=None):
        v = dict.get(self, k, d)
        return Cfg(v) if isinstance(v, dict) else v


def reference_classes():
    from transformers import BertTokenizer

    src = open(f"{REF}/antmmf/datasets/processors/text_processors.py").read()
    body = src[src.index('@registry.register_processor("masked_token")'):src.index('@registry.register_processor("masked_roberta_tokenizer")')]
    body = body[:body.index('@registry.register_processor("masked_layoutlm_tokenizer")')] + body[body.index('@registry.register_processor("masked_bert_tokenizer")'):]

    class BaseProcessor:
        def __init__(self, config, *a, **k):
            self.preprocessor = None

    class _Reg:
        @staticmethod
        def register_processor(name):
            return lambda c: c

    def build_tokenizer(cfg):
        params = dict(cfg["params"])
        return BertTokenizer(os.path.join(REF, "tests/data/vocabs", VOCABS[cfg["type"]]), **params)

    for name in ("antmmf", "antmmf.datasets", "antmmf.datasets.build", "antmmf.utils", "antmmf.utils.text_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["antmmf.datasets.build"].build_tokenizer = build_tokenizer
    tu_src = open(f"{REF}/antmmf/utils/text_utils.py").read()
    ns_tu = {}
    exec(tu_src[tu_src.index("def _is_chinese_char"):tu_src.index("def generate_ngrams")], ns_tu)
    sys.modules["antmmf.utils.text_utils"].is_chinese = ns_tu["is_chinese"]
    ns = dict(registry=_Reg, BaseProcessor=BaseProcessor, torch=torch, random=random, json=json, CLS_TOKEN_STR="[CLS]", SEP_TOKEN_STR="[SEP]",
              TEXT_MODALITY="text", CLS_ID_STR="cls_id", SEP_ID_STR="sep_id", LM_LABEL_IDS_STR="lm_label_ids")
    exec(body, ns)
    return ns["MaskedTokenProcessor"], ns["MaskedBertTokenizer"]


def captions():
    caps = []
    for f in ("msrvtt_train.jsonl", "msrvtt_test.jsonl", "VATEX_CN.jsonl", "univl_video.jsonl", "univl_img.jsonl"):
        for line in open(f"{REF}/tests/data/video/{f}"):
            line = line.strip()
            if line:
                c = json.loads(line).get("caption")
                if isinstance(c, str):
                    caps.append(c)
    caps += ["", "   ", "A man's dog -- isn't it? (yes!) 3.5kg, naïve café ÀÉÎ", "unaffable supercalifragilisticexpialidocious " + "z" * 120,
             "word " * 60, "[MASK] is a [SEP] token [CLS] [PAD] [UNK] inside", "混合 English 与中文 text，带标点。还有１２３ｆｕｌｌ",
             "tab\tnew\nline\r\x00nul�​zw ls", "ΣΟΦΟΣ İstanbul ǅ ß", "é vs é 寧", "\U0001F600 emoji \U00020000 ext"]
    return caps


def rows_of(out):
    r = {k: out[k].tolist() for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids")}
    r["tokens"], r["source_len"] = out["tokens"], out["source_len"]
    for k in ("cls_id", "sep_id"):
        if k in out:
            r[k] = out[k]
    return r


def main():
    MaskedToken, MaskedBert = reference_classes()
    caps = captions()
    cases = []

    def run(name, cls, params, items, seed=None, probability=None):
        proc = cls(Cfg(params))
        if seed is not None:
            random.seed(seed); torch.manual_seed(seed)
        outs = []
        for it in items:
            try:
                outs.append(rows_of(proc(dict(it)) if probability is None else proc(dict(it), probability=probability)))
            except AssertionError:   # a truncated PAIR is one token too long for the reference's own length assertion (content_len = max - 2, three specials)
                outs.append({"raises": "AssertionError"})
        cases.append(dict(name=name, cls=cls.__name__, params=params, items=items, seed=seed, probability=probability, expect=outs))

    tk = lambda v, lower: {"type": v, "params": {"model_type": "bert", "do_lower_case": lower, "pretrained": False}}
    text_items = [{"text": c} for c in caps]
    run("vtp_caption_uncased_30", MaskedBert, dict(max_seq_length=30, mask_probability=0, trim_start_token=False, tokenizer_config=tk("uncased", True)), text_items)
    run("vtp_caption_chinese_30", MaskedBert, dict(max_seq_length=30, mask_probability=0, tokenizer_config=tk("chinese", False)), text_items)
    run("caption_uncased_77_tokens_field", MaskedBert, dict(max_seq_length=77, tokenizer_config=tk("uncased", True)),
        [{"tokens": c.split()} for c in caps if c.strip()])
    run("trim_start_12", MaskedBert, dict(max_seq_length=12, trim_start_token=True, tokenizer_config=tk("uncased", True)), text_items)
    run("mlm_uncased_seeded", MaskedBert, dict(max_seq_length=40, mask_probability=0.3, whole_word_masking=True, tokenizer_config=tk("uncased", True)), text_items, seed=1234)
    run("mlm_chinese_only_seeded", MaskedBert, dict(max_seq_length=40, mask_probability=0.4, random_mask_chinese=True, tokenizer_config=tk("chinese", False)), text_items, seed=77)
    run("random_truncate_seeded", MaskedBert, dict(max_seq_length=10, random_truncate=True, tokenizer_config=tk("uncased", True)), text_items, seed=5)
    run("probability_override", MaskedBert, dict(max_seq_length=30, mask_probability=0.5, tokenizer_config=tk("uncased", True)), text_items[:8], seed=3, probability=0.0)
    pairs = [{"text_a": a, "text_b": b, "is_correct": i % 2} for i, (a, b) in enumerate(zip(caps, caps[3:] + caps[:3]))]
    run("pair_uncased_24", MaskedToken, dict(max_seq_length=24, mask_probability=0.0, tokenizer_config=tk("uncased", True)), pairs)
    run("pair_mlm_seeded", MaskedToken, dict(max_length=32, mask_probability=0.15, tokenizer_config=tk("uncased", True)), pairs, seed=9)
    with open(os.path.join(HERE, "text_processors.json"), "w", encoding="utf-8") as f:
        json.dump(dict(vocabs=VOCABS, cases=cases), f, ensure_ascii=False, separators=(",", ":"))
    print("cases", len(cases), "rows", sum(len(c["expect"]) for c in cases), "raising", sum("raises" in r for c in cases for r in c["expect"]))


def main_glm():
    """The M2 tokenizer call of prj/M2_Encoder/m2_encoder.py:39-45 through the reference's GLMChineseTokenizer (loaded from its file, with the
    reference's sp.model, which stays where it is: the test reads it from the reference checkout and skips without it)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_tokenization_glm", f"{REF}/prj/M2_Encoder/vlmo/tokenizer/tokenization_glm.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    tok = m.GLMChineseTokenizer.from_pretrained(f"{REF}/prj/M2_Encoder/vlmo/tokenizer")
    texts = captions() + ["a photo of   a cat [MASK] <|endoftext|> x", "两个  空格   三个    四个", "很长的句子" * 40, "[CLS][UNK] 开头", "x" * 300,
                          "杭州西湖的日落，游客在断桥上拍照。", "A dog runs on the beach; 一只狗在沙滩上奔跑!"]
    cases = []
    for L in (12, 52, 77):
        enc = tok(texts, padding="max_length", truncation=True, max_length=L, return_special_tokens_mask=True)
        cases.append(dict(max_length=L, input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]))
    pieces = [tok.tokenize(t) for t in texts]
    with open(os.path.join(HERE, "m2_tokenizer.json"), "w", encoding="utf-8") as f:
        json.dump(dict(texts=texts, cases=cases, pieces=pieces, ids=dict(cls=tok.cls_token_id, eos=tok.eos_token_id, pad=tok.pad_token_id, mask=tok.mask_token_id, unk=tok.unk_token_id), size=len(tok)),
                  f, ensure_ascii=False, separators=(",", ":"))
    print("glm texts", len(texts))


if __name__ == "__main__":
    main()
    main_glm()
