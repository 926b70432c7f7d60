"""roi_univl (base_vtp): importing this package registers the `univl` model and the CLIP-style encoders, as the
reference's prj/base_vtp/run.py:11 `import roi_univl` does."""
from .univl.model import clip_text_encoder, clip_visual_encoder, univl_model  # noqa: F401
