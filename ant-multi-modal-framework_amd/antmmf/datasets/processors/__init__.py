from .processors import BaseProcessor, Processor  # noqa: F401
from . import text_processors  # noqa: F401  (registers simple_sentence / masked_token / masked_bert_tokenizer)
from . import image_processors  # noqa: F401  (registers custom_transforms)
