"""Caption -> (input_ids, input_mask, segment_ids, lm_label_ids) on the host: the processors in front of the text tower
(reference antmmf/datasets/processors/text_processors.py: `simple_sentence` :558-574, `masked_token` :600-923,
`masked_bert_tokenizer` :1047-1080).  Same registry names, config keys, output keys, truncation rule and -- for
`mask_probability > 0` -- the same consumption order of `random` / `torch.randint`, so a seeded run masks the same positions.
Layout of one row: [CLS] a ... [SEP] (b ... [SEP]) then PAD (id 0) to `max_seq_length`; mask 1 over the real tokens; segment 1 over
the second sentence and its [SEP]; labels -1 except at masked positions.  The word-piece step is datasets/tokenization.py.
The SNP-S3 `intra_VTM.IW_MLM` important-word masking (:602-625,728-776) needs side files the reference does not ship: refused."""
import random
import re

import torch

from antmmf.common.registry import registry

from .processors import BaseProcessor, _get

TEXT_MODALITY, CLS_ID_STR, SEP_ID_STR, LM_LABEL_IDS_STR = "text", "cls_id", "sep_id", "lm_label_ids"   # antmmf/common/constants.py:29-47
_SENTENCE_SPLIT = re.compile(r"(\W+)")


def simple_tokenize(sentence, keep=("'s",), remove=(",", "?")):
    """antmmf/utils/text_utils.py:285-295: lower, keep "'s" as its own token, drop ',' and '?', split on non-word runs."""
    sentence = sentence.lower()
    for tok in keep:
        sentence = sentence.replace(tok, " " + tok)
    for tok in remove:
        sentence = sentence.replace(tok, "")
    return [t.strip() for t in _SENTENCE_SPLIT.split(sentence) if len(t.strip()) > 0]


@registry.register_processor("simple_sentence")
class SimpleSentenceProcessor(BaseProcessor):
    def __init__(self, *args, **kwargs):
        self.tokenizer = simple_tokenize

    def __call__(self, item, *args, **kwargs):
        return {"text": self.tokenizer(item["text"], *args, **kwargs)}


def _is_chinese(token):
    """antmmf/utils/text_utils.py:238-244: every character in one of the CJK ideograph blocks (vacuously true for "")."""
    from antmmf.datasets.tokenization import _CJK_RANGES

    return all(any(lo <= ord(ch) <= hi for lo, hi in _CJK_RANGES) for ch in token)


@registry.register_processor("masked_token")
class MaskedTokenProcessor(BaseProcessor):
    _CLS_TOKEN, _SEP_TOKEN, _MASK_TOKEN, _PAD_TOKEN_ID = "[CLS]", "[SEP]", "[MASK]", 0

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)
        from antmmf.datasets.build import build_tokenizer

        self._tokenizer = build_tokenizer(_get(config, "tokenizer_config"))
        self._max_seq_length = _get(config, "max_length", None) if _get(config, "max_length", None) is not None else _get(config, "max_seq_length", None)
        assert self._max_seq_length is not None, "max_seq_length is not set in config"
        self._probability = _get(config, "mask_probability", 0.15)
        self._trim_start_token = _get(config, "trim_start_token", False)
        self._random_mask_chinese = _get(config, "random_mask_chinese", False)
        self._random_truncate = _get(config, "random_truncate", False)
        self._wwm = _get(config, "whole_word_masking", False)
        intra = _get(config, "intra_VTM", False)
        if intra and _get(intra, "IW_MLM", False):
            raise NotImplementedError("masked_token: intra_VTM.IW_MLM (SNP-S3 important-word masking) needs word-rank / lemma files that are "
                                      "not part of the contrastive path")

    def get_vocab_size(self):
        return len(self._tokenizer)

    def tokenizer(self):
        return self._tokenizer

    # ---- masking (BERT 80 / 10 / 10); one random.random() per token, one torch.randint per random replacement
    def _random_word(self, tokens, probability=0.15):
        labels = []
        for idx, token in enumerate(tokens):
            draw = random.random()
            if self._random_mask_chinese and not _is_chinese(token):
                labels.append(-1)
                continue
            if draw >= probability:
                labels.append(-1)
                continue
            draw /= probability
            if draw < 0.8:
                tokens[idx] = self._MASK_TOKEN
            elif draw < 0.9:
                tokens[idx] = self._tokenizer.convert_ids_to_tokens(torch.randint(len(self._tokenizer), (1,), dtype=torch.long))[0]
            labels.append(self._tokenizer.convert_tokens_to_ids(token))
        return tokens, labels

    def _whole_word_masking(self, tokens, labels):
        """A "##" continuation piece whose word start was chosen is masked with it."""
        out_tokens, out_labels = tokens[:], labels[:]
        for i in range(1, len(tokens)):
            if not tokens[i].startswith("##"):
                continue
            head = i - 1
            while head >= 0 and tokens[head].startswith("##"):
                head -= 1
            if head >= 0 and labels[head] != -1:
                out_labels[i] = self._tokenizer.convert_tokens_to_ids(tokens[i])
                out_tokens[i] = self._MASK_TOKEN
        return out_tokens, out_labels

    def _truncate_tokens(self, tokens, max_length, random_truncate=True):
        if random_truncate:   # a random window (LayoutLMv2 3.2)
            s = random.randint(0, max(len(tokens) - max_length, 0))
            return tokens[s:s + max_length]
        return tokens[:min(max_length, len(tokens))]

    def _truncate_seq_pair(self, tokens_a, tokens_b, max_length):
        if tokens_b is None:
            return self._truncate_tokens(tokens_a, max_length, random_truncate=self._random_truncate), []
        while len(tokens_a) + len(tokens_b) > max_length:   # always shorten the longer one, from its end
            (tokens_a if len(tokens_a) > len(tokens_b) else tokens_b).pop()
        return tokens_a, tokens_b

    def _convert_to_indices(self, tokens_a, tokens_b=None, probability=0.15):
        tokens_a, label_a = self._random_word(tokens_a, probability=probability)
        if self._wwm:
            tokens_a, label_a = self._whole_word_masking(tokens_a, label_a)
        if self._trim_start_token:
            tokens, segment_ids, labels = [], [], []
        else:
            tokens, segment_ids, labels = [self._CLS_TOKEN], [0], [-1]
        tokens = tokens + tokens_a + [self._SEP_TOKEN]
        segment_ids = segment_ids + [0] * (len(tokens_a) + 1)
        if tokens_b:
            tokens_b, label_b = self._random_word(tokens_b, probability=probability)
            labels = labels + label_a + [-1] + label_b + [-1]
            tokens = tokens + tokens_b + [self._SEP_TOKEN]
            segment_ids = segment_ids + [1] * (len(tokens_b) + 1)
        else:
            labels = labels + label_a + [-1]
        ids = self._tokenizer.convert_tokens_to_ids(tokens)
        source_len, L = len(ids), self._max_seq_length
        assert source_len <= L, (source_len, L)
        pad = L - source_len
        return {
            "input_ids": torch.tensor(ids + [self._PAD_TOKEN_ID] * pad, dtype=torch.long),
            "input_mask": torch.tensor([1] * source_len + [0] * pad, dtype=torch.long),
            "segment_ids": torch.tensor(segment_ids + [0] * pad, dtype=torch.long),
            "lm_label_ids": torch.tensor(labels + [-1] * pad, dtype=torch.long),
            "tokens": tokens,
            "source_len": source_len,
        }

    def _content_len(self):
        return self._max_seq_length - (1 if self._trim_start_token else 2)

    def __call__(self, item, probability=None):
        text_a = item["text_a"] if "text_a" in item else item["text"]
        text_b = item.get("text_b", None)
        tokens_a = self._tokenizer.tokenize(text_a)
        tokens_b = self._tokenizer.tokenize(text_b) if text_b else None
        tokens_a, tokens_b = self._truncate_seq_pair(tokens_a, tokens_b, self._content_len())
        out = self._convert_to_indices(tokens_a, tokens_b, probability=probability if probability is not None else self._probability)
        if "is_correct" in item:
            out["is_correct"] = torch.tensor(item["is_correct"], dtype=torch.long)
        return out

    def batch(self, texts, probability=None, device=None):
        """Whole caption batch at once -> {input_ids, input_mask, segment_ids} as [B, L] tensors (one host -> device copy per key when
        `device` is given).  Same rows as calling the processor per caption (one text field per caption; with mask probability > 0 the draws are made
        caption by caption in order)."""
        rows = [self({"text": t}, probability=probability) for t in texts]
        out = {k: torch.stack([r[k] for r in rows]) for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids")}
        if device is not None:
            out = {k: v.to(device, non_blocking=True) for k, v in out.items()}
        return out


@registry.register_processor("masked_bert_tokenizer")
class MaskedBertTokenizer(MaskedTokenProcessor):
    """The caption processor of every *_vtp yml: mask probability 0 unless configured, one text field ("text" or pre-split "tokens")."""

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)
        self._probability = _get(config, "mask_probability", 0)
        self._trim_start_token = _get(config, "trim_start_token", False)

    def __call__(self, item, probability=None):
        text_a = item["text"] if "text" in item else " ".join(item["tokens"])
        tokens_a = self._tokenizer.tokenize(text_a)
        tokens_a, _ = self._truncate_seq_pair(tokens_a, None, self._content_len())
        out = self._convert_to_indices(tokens_a, None, probability=probability if probability is not None else self._probability)
        out[TEXT_MODALITY] = out["tokens"]
        out[CLS_ID_STR] = self._tokenizer.cls_token_id
        out[SEP_ID_STR] = self._tokenizer.sep_token_id
        out[LM_LABEL_IDS_STR] = out["lm_label_ids"]
        return out
