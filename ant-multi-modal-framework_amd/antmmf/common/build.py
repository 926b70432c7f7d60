"""build_config (reference: antmmf/common/build.py:8-34): YAML + defaults + `k v` overrides -> registry["config"]."""
from antmmf.common.configuration import Configuration
from antmmf.common.registry import registry


def build_config(config_yaml, config_override=None, opts_override=None, root=None):
    cfg = Configuration.from_file(config_yaml, with_defaults=True, root=root)
    if config_override:
        cfg.update_nested(config_override)
    cfg.override_with_cmd_opts(opts_override or [])
    registry.register("config", cfg)
    return cfg
