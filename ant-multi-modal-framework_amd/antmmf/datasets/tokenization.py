"""BERT text -> word-piece ids on the host (the tokenisation step in front of the text tower).

The reference builds its tokenizer with HuggingFace `AutoTokenizer.from_pretrained(<bert dir>)`
(antmmf/datasets/build.py:64-120), i.e. the arithmetic lives in a third-party dependency (`transformers`, BertTokenizer:
"basic" tokenisation + greedy longest-match word pieces, the algorithm published with BERT; `transformers>=4.17.0` in
requirements.txt:17 resolves AutoTokenizer to the `tokenizers`-backed class, whose character classes differ from the pure-Python
one in corner cases -- unassigned code points are kept, U+2028 / U+2029 are spaces, lower-casing is per character -- and are the
ones restated here; transformers 5.15.0 / tokenizers in this image).  This file restates that algorithm
as one class with the surface the reference's processors use (text_processors.py:621-1076): `tokenize`,
`convert_tokens_to_ids`, `convert_ids_to_tokens`, `__len__`, `cls_token_id`, `sep_token_id`, `pad_token_id`, `mask_token`.
Pinned bit-exactly against transformers' BertTokenizer on the reference's own vocabularies
(tests/data/vocabs/bert-base-{uncased_30522,chinese_21128}_vocab.txt): tests/golden/tokenizer_bert.json, tests/test_tokenizer.py.

Design: the per-character classification (drop / space / CJK / punctuation / keep) is a table lookup built once per process for
the BMP and cached per code point above it; the word-piece search runs on a dict keyed by the piece text with a per-vocabulary
bound on the piece length, so a 100-character word costs at most 100 x max_piece_len probes instead of 100^2 slices.
"""
import collections
import os
import unicodedata

_SPECIALS = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")

# character classes
_DROP, _SPACE, _CJK, _PUNCT, _KEEP = 0, 1, 2, 3, 4
_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF),
               (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _classify(cp):
    ch = chr(cp)
    if ch in " \t\n\r":
        return _SPACE
    if cp == 0 or cp == 0xFFFD:
        return _DROP
    cat = unicodedata.category(ch)
    if cat in ("Cc", "Cf", "Co"):   # control / format / private-use characters are removed (unassigned code points are kept)
        return _DROP
    if cat == "Zs" or cp == 0x2028 or cp == 0x2029:   # Unicode White_Space: the space separators + line / paragraph separator
        return _SPACE
    for lo, hi in _CJK_RANGES:
        if lo <= cp <= hi:
            return _CJK
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126) or cat[0] == "P":
        return _PUNCT
    return _KEEP


_CLASS_CACHE = {}


def _char_class(ch):
    c = _CLASS_CACHE.get(ch)
    if c is None:
        c = _CLASS_CACHE[ch] = _classify(ord(ch))
    return c


def load_vocab(vocab_file):
    """One token per line, id = line number (trailing newline stripped, nothing else)."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as f:
        for i, line in enumerate(f.readlines()):
            vocab[line.rstrip("\n")] = i
    return vocab


class BertWordPieceTokenizer:
    def __init__(self, vocab_file, do_lower_case=True, tokenize_chinese_chars=True, strip_accents=None, unk_token="[UNK]",
                 sep_token="[SEP]", pad_token="[PAD]", cls_token="[CLS]", mask_token="[MASK]", max_input_chars_per_word=100, **_ignored):
        if not os.path.isfile(vocab_file):
            raise ValueError(f"BertWordPieceTokenizer: no vocabulary file at {vocab_file!r}")
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens = {i: t for t, i in self.vocab.items()}   # a vocabulary file may repeat a token: the last line wins, ids keep gaps
        self.do_lower_case = bool(do_lower_case)
        self.tokenize_chinese_chars = bool(tokenize_chinese_chars)
        self.strip_accents = strip_accents
        self.unk_token, self.sep_token, self.pad_token, self.cls_token, self.mask_token = unk_token, sep_token, pad_token, cls_token, mask_token
        self.max_input_chars_per_word = max_input_chars_per_word
        self.all_special_tokens = [unk_token, sep_token, pad_token, cls_token, mask_token]
        self._special_set = set(self.all_special_tokens)
        self._max_piece = max((len(t) for t in self.vocab), default=1)
        self._unk_id = self.vocab.get(unk_token)

    # ---- surface used by the processors
    def __len__(self):
        return len(self.vocab)

    @property
    def vocab_size(self):
        return len(self.vocab)

    cls_token_id = property(lambda self: self.vocab.get(self.cls_token))
    sep_token_id = property(lambda self: self.vocab.get(self.sep_token))
    pad_token_id = property(lambda self: self.vocab.get(self.pad_token))
    mask_token_id = property(lambda self: self.vocab.get(self.mask_token))
    unk_token_id = property(lambda self: self._unk_id)

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.vocab.get(tokens, self._unk_id)
        get, unk = self.vocab.get, self._unk_id
        return [get(t, unk) for t in tokens]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.ids_to_tokens.get(ids, self.unk_token)
        return [self.ids_to_tokens.get(int(i), self.unk_token) for i in ids]   # a tensor of ids (text_processors.py:717-719) or a list

    # ---- text -> tokens
    def _split_specials(self, text):
        """Special tokens written out in the text ("[SEP]", "[MASK]" ...) survive as single tokens: split the text around them
        (longest match first at every position, as a trie over the special tokens would)."""
        specials = sorted(self._special_set, key=len, reverse=True)
        parts, start, i, n = [], 0, 0, len(text)
        while i < n:
            hit = None
            if text[i] == "[" or any(s[0] == text[i] for s in specials):
                for s in specials:
                    if text.startswith(s, i):
                        hit = s
                        break
            if hit is None:
                i += 1
                continue
            if i > start:
                parts.append((text[start:i], False))
            parts.append((hit, True))
            i += len(hit)
            start = i
        if start < n:
            parts.append((text[start:], False))
        return parts

    def _basic(self, text):
        """clean -> space out CJK characters -> whitespace words -> accent strip / lower -> split at punctuation."""
        buf = []
        for ch in text:
            c = _char_class(ch)
            if c == _DROP:
                continue
            if c == _SPACE:
                buf.append(" ")
            elif c == _CJK and self.tokenize_chinese_chars:
                buf.append(" "); buf.append(ch); buf.append(" ")
            else:
                buf.append(ch)
        text = "".join(buf)   # (no NFC pass: the `tokenizers` normalizer has none; the pure-Python BertTokenizer of transformers 4.x does)
        out = []
        strip = self.strip_accents if self.strip_accents is not None else self.do_lower_case
        for word in text.split():
            if strip:   # accents first, then case: the order of the `tokenizers` BertNormalizer
                word = "".join(ch for ch in unicodedata.normalize("NFD", word) if unicodedata.category(ch) != "Mn")
            if self.do_lower_case:
                word = "".join(ch.lower() for ch in word)   # per character: no final-sigma context rule
            cur = []
            for ch in word:
                if _char_class(ch) == _PUNCT:
                    if cur:
                        out.append("".join(cur)); cur = []
                    out.append(ch)
                else:
                    cur.append(ch)
            if cur:
                out.append("".join(cur))
        # a word can contain characters that only become whitespace / empty after the steps above
        return " ".join(out).split()

    def _wordpiece(self, word, out):
        n = len(word)
        if n > self.max_input_chars_per_word:
            out.append(self.unk_token)
            return
        vocab, maxp = self.vocab, self._max_piece
        pieces, start = [], 0
        while start < n:
            end = min(n, start + (maxp if start == 0 else maxp - 2))
            hit = None
            while end > start:
                piece = word[start:end] if start == 0 else "##" + word[start:end]
                if piece in vocab:
                    hit = piece
                    break
                end -= 1
            if hit is None:
                out.append(self.unk_token)
                return
            pieces.append(hit)
            start = end
        out.extend(pieces)

    def tokenize(self, text):
        out = []
        for chunk, is_special in self._split_specials(text):
            if is_special:
                out.append(chunk)
                continue
            for word in self._basic(chunk):
                self._wordpiece(word, out)
        return out

    def encode(self, text, max_length=None):
        """[CLS] tokens [SEP] -> ids (convenience; the processors assemble the sequence themselves)."""
        toks = self.tokenize(text)
        if max_length is not None:
            toks = toks[:max_length - 2]
        return self.convert_tokens_to_ids([self.cls_token] + toks + [self.sep_token])
