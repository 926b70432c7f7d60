"""Processor plumbing of the reference (antmmf/datasets/processors/processors.py:85-190): `BaseProcessor` (optional nested
`preprocessor`), and `Processor(config)` = look the class up in the registry by `config.type` and build it from `config.params`."""
from antmmf.common.registry import registry


def _get(cfg, key, default=None):
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class BaseProcessor:
    def __init__(self, config, *args, **kwargs):
        self.config = config
        self.preprocessor = None
        pre = _get(config, "preprocessor", None)
        if pre is not None:
            self.preprocessor = Processor(pre, *args, **kwargs)

    def __call__(self, item, *args, **kwargs):
        return item


class Processor:
    """Wrapper built from {type: <registered name>, params: {...}}; calls and attribute reads go to the wrapped processor."""

    def __init__(self, config, *args, **kwargs):
        ptype = _get(config, "type", None)
        if ptype is None:
            raise AttributeError("Config must have 'type' attribute to specify type of processor")
        cls = registry.get_processor_class(ptype)
        if cls is None:
            raise ValueError(f"No processor named {ptype} is defined.")
        params = _get(config, "params", None)
        self.processor = cls(params if params is not None else {}, *args, **kwargs)

    def __call__(self, item, *args, **kwargs):
        return self.processor(item, *args, **kwargs)

    def __getattr__(self, name):
        if name == "processor":
            raise AttributeError(name)
        return getattr(self.processor, name)
