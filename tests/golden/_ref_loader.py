"""Loader that imports the hot-path files of the *reference* AntMMF checkout, unmodified.

Build-container tool only.  It exists so that `make_golden.py` can run the
reference implementation on seeded inputs and write small input/output tensors under
`tests/golden/`.  Nothing in `tests/` that runs on the GPU box imports this file, and no reference
source text is stored in the repo: only tensors produced by executing it.
"""
import contextlib
import importlib
import os
import sys
import types

import torch

REF = os.environ.get("ANTMMF_REFERENCE", "/root/reference")


class AttrDict(dict):
    """attribute-style dict standing in for antmmf.common.Configuration inside the reference."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def get(self, k, d=None):
        v = dict.get(self, k, d)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


_loaded = {}


def load_antmmf_core():
    if "core" in _loaded:
        return _loaded["core"]
    import transformers  # noqa: F401  (must be imported before any stub exists)

    for n, p in [
        ("antmmf", "antmmf"),
        ("antmmf.common", "antmmf/common"),
        ("antmmf.utils", "antmmf/utils"),
        ("antmmf.modules", "antmmf/modules"),
        ("antmmf.modules.vision", "antmmf/modules/vision"),
        ("antmmf.modules.vision.backbone", "antmmf/modules/vision/backbone"),
        ("antmmf.modules.vision.backbone.clip", "antmmf/modules/vision/backbone/clip"),
    ]:
        _pkg(n, f"{REF}/{p}")
    sys.modules["antmmf.common"].configurable = lambda f: f
    sys.modules["antmmf.common"].Configuration = AttrDict
    g = types.ModuleType("antmmf.utils.general")
    g.nullcontext = contextlib.nullcontext
    g.get_package_version = lambda n: torch.__version__.split("+")[0]
    g.check_required_keys = lambda d, required_keys=[]: all(k in d for k in required_keys)
    sys.modules["antmmf.utils.general"] = g
    mr = importlib.import_module("antmmf.modules.module_registry")
    enc = types.ModuleType("antmmf.modules.encoders")

    class TextEncoder(mr.ModuleRegistry):
        def __init__(s, config, *a, **k):
            super().__init__(config.type, *a, **config.get("params", {}), **k)

    class VisualEncoder(mr.ModuleRegistry):
        def __init__(s, config, *a, **k):
            super().__init__(config.type, *a, **k, **config.get("params", {}))

    enc.TextEncoder, enc.VisualEncoder = TextEncoder, VisualEncoder
    sys.modules["antmmf.modules.encoders"] = enc
    out = dict(
        vit=importlib.import_module("antmmf.modules.vision.backbone.clip.model"),
        bert=importlib.import_module("antmmf.modules.vision.backbone.clip.modeling_bert"),
        bert_cfg=importlib.import_module("antmmf.modules.vision.backbone.clip.configuration_bert"),
        du=importlib.import_module("antmmf.utils.distributed_utils"),
    )
    _loaded["core"] = out
    return out


def load_vtp(project="base_vtp"):
    """Returns the reference's univl retrieval modules of prj/<project> (base_vtp | dmae_vtp | cnvid_vtp)."""
    core = load_antmmf_core()
    for k in [k for k in sys.modules if k.startswith("roi_univl")]:
        del sys.modules[k]
    P = f"{REF}/prj/{project}/roi_univl"
    for n, p in [("roi_univl", P), ("roi_univl.univl", P + "/univl"), ("roi_univl.univl.model", P + "/univl/model")]:
        _pkg(n, p)
    if project == "cnvid_vtp":   # ships no CLIP-style encoders (its configs pair the model with Video-Swin): the tiny test towers come from base_vtp's files
        sys.modules["roi_univl.univl.model"].__path__.append(f"{REF}/prj/base_vtp/roi_univl/univl/model")
    src = open(P + "/univl/model/univl_base.py").read()
    ub = types.ModuleType("roi_univl.univl.model.univl_base")
    exec("import torch\n" + src[src.index("def split_encoder_output"):src.index("class UniVlBase")], ub.__dict__)
    sys.modules["roi_univl.univl.model.univl_base"] = ub
    out = dict(core)
    out["vis_enc"] = importlib.import_module("roi_univl.univl.model.clip_visual_encoder")
    out["txt_enc"] = importlib.import_module("roi_univl.univl.model.clip_text_encoder")
    out["ret"] = importlib.import_module("roi_univl.univl.model.univl_video_ret")
    out["moco"] = importlib.import_module("roi_univl.univl.model.moco_utils")
    if project == "dmae_vtp":
        out["dmae"] = importlib.import_module("roi_univl.univl.model.dmae_utils")
    return out


def load_m2():
    """Returns the reference's M2_Encoder VLMo class + default config dict."""
    if "m2" in _loaded:
        return _loaded["m2"]
    import transformers  # noqa: F401
    import torch.nn as nn

    P = f"{REF}/prj/M2_Encoder"
    if P not in sys.path:
        sys.path.insert(0, P)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    ident = lambda x, *a, **k: x
    stub("fairscale")
    stub("fairscale.nn", checkpoint_wrapper=ident, wrap=ident)

    class LightningModule(nn.Module):
        def save_hyperparameters(self):
            import inspect

            frame = inspect.currentframe().f_back
            self.hparams = types.SimpleNamespace(config=frame.f_locals["config"])

    stub("pytorch_lightning", LightningModule=LightningModule)
    stub("pytorch_lightning.utilities")
    stub("pytorch_lightning.utilities.distributed", rank_zero_info=lambda *a, **k: None)

    def drop_path(x, drop_prob=0.0, training=False):
        assert drop_prob == 0.0 or not training
        return x

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    stub("timm")
    stub("timm.models", create_model=None)
    stub("timm.models.layers", drop_path=drop_path, trunc_normal_=trunc_normal_)

    class _T:
        def __init__(self, *a, **k):
            pass

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
            s = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
            return (x - m) / s

    stub("torchvision")
    stub("torchvision.transforms", Compose=Compose, Normalize=Normalize, Resize=_T, ToTensor=_T, RandomResizedCrop=_T,
         RandomHorizontalFlip=_T, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    stub("cv2")

    class Experiment:
        def __init__(self, *a, **k):
            pass

        def config(self, f):
            return f

        def named_config(self, f):
            return f

    stub("sacred", Experiment=Experiment)
    src = open(P + "/vlmo/config.py").read()
    importlib.import_module("vlmo.modules")
    vm = importlib.import_module("vlmo.modules.vlmo_module")
    cfgmod = importlib.import_module("vlmo.config")
    out = dict(VLMo=vm.VLMo, vm=vm, cfgmod=cfgmod, cfg_src=src, root=P)
    _loaded["m2"] = out
    return out
