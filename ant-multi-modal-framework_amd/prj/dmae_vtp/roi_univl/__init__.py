"""roi_univl (dmae_vtp): the base_vtp package plus the DMAE stage-3 head.  Modules that DMAE does not override (encoders, towers,
MoCo, the retrieval model itself -- the reference's dmae_vtp copies differ from base_vtp only by stage 3) resolve to base_vtp's
files through the extended package path; `univl.model.dmae_utils` lives here.  Put prj/dmae_vtp on sys.path and `import roi_univl`
as the reference's prj/dmae_vtp/run.py does."""
import os

__path__.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "base_vtp", "roi_univl")))
from .univl.model import clip_text_encoder, clip_visual_encoder, univl_model  # noqa: E402,F401
