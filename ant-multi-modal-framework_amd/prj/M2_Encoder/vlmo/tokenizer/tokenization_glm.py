"""GLM Chinese tokenizer of the M2 encoder (reference prj/M2_Encoder/vlmo/tokenizer/tokenization_glm.py:208-295 on top of
transformers' PreTrainedTokenizer, called from prj/M2_Encoder/m2_encoder.py:39-45,74-80 as
`tokenizer(texts, padding="max_length", truncation=True, max_length=max_text_len)`).

The piece model is SentencePiece (third-party library `sentencepiece`, model file `sp.model` shipped with the reference's checkpoint
directory -- a data file like the weights).  What the reference adds around it, restated here without the transformers base class:
runs of 2..10 blanks become `<|blank_n|>` pieces before encoding; special tokens written out in the text survive as single tokens
(the text is split around them before the piece model sees it); a row is `[CLS] ids <|endoftext|>`; truncation drops ids from the
LEFT (`truncation_side = "left"`); padding to max_length uses the pad id (= `<|endoftext|>`) on the right with attention_mask 0.
Pinned against the reference class itself: tests/golden/m2_tokenizer.json (tests/golden/make_golden_text.py)."""
import json
import os


def encode_whitespaces(content):
    for i in range(10, 1, -1):
        content = content.replace(" " * i, f"<|blank_{i}|>")
    return content


def decode_whitespaces(content):
    for i in range(10, 1, -1):
        content = content.replace(f"<|blank_{i}|>", " " * i)
    return content


class GLMChineseTokenizer:
    vocab_files_names = {"vocab_file": "sp.model"}
    truncation_side = "left"

    def __init__(self, vocab_file, eos_token="<|endoftext|>", pad_token="<|endoftext|>", cls_token="[CLS]", mask_token="[MASK]",
                 unk_token="[UNK]", **_ignored):
        try:
            import sentencepiece as spm
        except ImportError as e:   # the reference's own dependency (prj/M2_Encoder/requirements.txt)
            raise ImportError("GLMChineseTokenizer needs the `sentencepiece` package") from e
        if not os.path.isfile(vocab_file):
            raise FileNotFoundError(f"GLMChineseTokenizer: no SentencePiece model at {vocab_file!r}")
        self.vocab_file = vocab_file
        self.sp_model = spm.SentencePieceProcessor()
        self.sp_model.Load(vocab_file)
        self.eos_token, self.pad_token, self.cls_token, self.mask_token, self.unk_token = eos_token, pad_token, cls_token, mask_token, unk_token
        seen, self.all_special_tokens = set(), []
        for t in (eos_token, pad_token, cls_token, mask_token, unk_token):
            if t not in seen:
                seen.add(t); self.all_special_tokens.append(t)
        self._specials_longest_first = sorted(self.all_special_tokens, key=len, reverse=True)

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """`path`: the directory holding sp.model (+ tokenizer_config.json), or the model file itself."""
        if os.path.isdir(path):
            cfg_file = os.path.join(path, "tokenizer_config.json")
            cfg = {}
            if os.path.isfile(cfg_file):
                with open(cfg_file) as f:
                    cfg = {k: v for k, v in json.load(f).items() if k.endswith("_token")}
            cfg.update(kwargs)
            return cls(os.path.join(path, cls.vocab_files_names["vocab_file"]), **cfg)
        return cls(path, **kwargs)

    # ---- vocabulary
    @property
    def vocab_size(self):
        return len(self.sp_model)

    def __len__(self):
        return len(self.sp_model)

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.sp_model.PieceToId(tokens)
        return [self.sp_model.PieceToId(t) for t in tokens]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.sp_model.IdToPiece(ids)
        return [self.sp_model.IdToPiece(int(i)) for i in ids]

    def convert_tokens_to_string(self, ids):
        return decode_whitespaces(self.sp_model.DecodeIds(list(ids)))

    cls_token_id = property(lambda self: self.sp_model.PieceToId(self.cls_token))
    eos_token_id = property(lambda self: self.sp_model.PieceToId(self.eos_token))
    pad_token_id = property(lambda self: self.sp_model.PieceToId(self.pad_token))
    mask_token_id = property(lambda self: self.sp_model.PieceToId(self.mask_token))
    unk_token_id = property(lambda self: self.sp_model.PieceToId(self.unk_token))

    # ---- text -> pieces
    def tokenize(self, text):
        out, start, i, n = [], 0, 0, len(text)

        def flush(chunk):
            if chunk:
                out.extend(self.sp_model.EncodeAsPieces(encode_whitespaces(chunk)))

        while i < n:
            hit = next((s for s in self._specials_longest_first if text.startswith(s, i)), None)
            if hit is None:
                i += 1
                continue
            flush(text[start:i])
            out.append(hit)
            i += len(hit)
            start = i
        flush(text[start:])
        return out

    def build_inputs_with_special_tokens(self, token_ids_0, token_ids_1=None):
        assert token_ids_1 is None
        return [self.cls_token_id] + list(token_ids_0) + [self.eos_token_id]

    def __call__(self, texts, padding=False, truncation=False, max_length=None, return_special_tokens_mask=False, return_tensors=None, **_ignored):
        single = isinstance(texts, str)
        rows_ids, rows_mask = [], []
        for text in ([texts] if single else texts):
            ids = self.convert_tokens_to_ids(self.tokenize(text))
            if truncation and max_length is not None and len(ids) + 2 > max_length:
                ids = ids[len(ids) + 2 - max_length:]   # left side
            ids = self.build_inputs_with_special_tokens(ids)
            mask = [1] * len(ids)
            if padding == "max_length" and max_length is not None and len(ids) < max_length:
                pad = max_length - len(ids)
                ids, mask = ids + [self.pad_token_id] * pad, mask + [0] * pad
            rows_ids.append(ids); rows_mask.append(mask)
        if padding is True or padding == "longest":
            L = max(len(r) for r in rows_ids)
            rows_mask = [m + [0] * (L - len(m)) for m in rows_mask]
            rows_ids = [r + [self.pad_token_id] * (L - len(r)) for r in rows_ids]
        if single:
            rows_ids, rows_mask = rows_ids[0], rows_mask[0]
        out = {"input_ids": rows_ids, "attention_mask": rows_mask}
        if return_tensors == "pt":
            import torch

            out = {k: torch.tensor(v, dtype=torch.long) for k, v in out.items()}
        return out


class GLMTokenizer:
    @classmethod
    def from_pretrained(cls, path, *inputs, **kwargs):
        return GLMChineseTokenizer.from_pretrained(path, **kwargs)
