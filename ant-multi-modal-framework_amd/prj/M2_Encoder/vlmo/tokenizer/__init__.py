from .tokenization_glm import GLMChineseTokenizer  # noqa: F401
