"""Encoder registry families (reference: antmmf/modules/encoders/text_encoder.py:22-29,
visual_encoder.py:34-50): `TextEncoder(config).module`, `VisualEncoder(config).module` with config.type = class
name and config.params = constructor kwargs."""
from antmmf.modules.module_registry import ModuleRegistry


def _params(config):
    p = config.get("params", {}) if hasattr(config, "get") else {}
    return dict(p) if p is not None else {}


class TextEncoder(ModuleRegistry):
    def __init__(self, config, *args, **kwargs):
        super().__init__(config["type"], *args, **_params(config), **kwargs)


class VisualEncoder(ModuleRegistry):
    def __init__(self, config, *args, **kwargs):
        super().__init__(config["type"], *args, **kwargs, **_params(config))


from . import text_encoder  # noqa: E402,F401  (registers PretrainedTransformerEncoder)
