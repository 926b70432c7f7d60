"""`build_tokenizer(config)` with the contract of the reference (antmmf/datasets/build.py:64-120): `config.type` names a pretrained model
directory under $PYTORCH_TRANSFORMERS_CACHE (holding `vocab.txt`) -- or, with `params.pretrained: false`, is itself the path of a
vocabulary file unless `params.vocab_path` gives one; `params.model_type` decides the tokenizer class.  The reference hands this to
HuggingFace AutoTokenizer; this build ships the BERT word-piece tokenizer itself (datasets/tokenization.py) and needs no network:
a vocabulary that cannot be found is an error, not a download."""
import os

from .tokenization import BertWordPieceTokenizer

BERT_PRETRAINED_MODELS_ENV_VAR = "PYTORCH_TRANSFORMERS_CACHE"   # antmmf/common/constants.py


def get_transformer_model_vocab_path(name):
    """antmmf/utils/general.py:413-445: absolute paths stay, names resolve under $PYTORCH_TRANSFORMERS_CACHE."""
    if os.path.isabs(name):
        return name
    return os.path.join(os.environ.get(BERT_PRETRAINED_MODELS_ENV_VAR, ""), name)


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def build_tokenizer(config, *args, **kwargs):
    params = _get(config, "params", None) or {}
    params = dict(params) if not isinstance(params, dict) else dict(params)
    name = _get(config, "type")
    if name is None:
        raise ValueError("build_tokenizer: tokenizer_config.type is required")
    model_type = params.pop("model_type", "bert")
    pretrained = params.pop("pretrained", True)
    vocab_path = params.pop("vocab_path", None)
    if model_type not in (None, "bert"):
        raise NotImplementedError(f"build_tokenizer: model_type {model_type!r} -- only the BERT word-piece tokenizer is on the contrastive "
                                  "video / image-text path (SURVEY.md 8(f4)); roberta / sentencepiece families are out of scope")
    if pretrained:
        root = get_transformer_model_vocab_path(name)
        vocab_file = os.path.join(root, "vocab.txt") if os.path.isdir(root) else root
        if not os.path.isfile(vocab_file):
            # the reference falls back to downloading `name`; there is no network here -- say what is missing
            raise FileNotFoundError(f"build_tokenizer: no vocabulary for {name!r}: expected {os.path.join(root, 'vocab.txt')} "
                                    f"(set ${BERT_PRETRAINED_MODELS_ENV_VAR} to the directory holding {name}/vocab.txt)")
        cfg_json = os.path.join(root, "tokenizer_config.json") if os.path.isdir(root) else None
        if cfg_json and os.path.isfile(cfg_json) and "do_lower_case" not in params:
            import json

            with open(cfg_json) as f:
                params.setdefault("do_lower_case", json.load(f).get("do_lower_case", True))
        if "do_lower_case" not in params:
            # what transformers' hard-coded init table gives the stock model names (bert-base-chinese: False; *-uncased: True; *-cased: False)
            base = os.path.basename(os.path.normpath(name))
            params["do_lower_case"] = not (base == "bert-base-chinese" or (base.endswith("-cased") or "-cased-" in base))
    else:
        vocab_file = os.path.abspath(name) if vocab_path is None else os.path.abspath(os.path.expanduser(vocab_path))
    return BertWordPieceTokenizer(vocab_file, **params)
