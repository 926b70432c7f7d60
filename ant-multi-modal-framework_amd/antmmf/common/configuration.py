"""Configuration: nested attribute-style mapping built from YAML with `includes:`, environment-variable
expansion, dotted command-line overrides and freeze / defrost (behaviour of the reference's
antmmf/common/configuration.py:106-139,240-491, without the omegaconf dependency).

    cfg = Configuration.from_file("prj/base_vtp/configs/.../msr_vtt.yml")   # includes merged depth-first
    cfg.override_with_cmd_opts(["training_parameters.batch_size", "8"])     # literal_eval'ed values
    cfg.model_attributes.univl.hidden_size
"""
import ast
import collections.abc
import copy
import os

import yaml

_DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "defaults", "configs", "base.yml")


def nested_dict_update(dst, src):
    for k, v in src.items():
        if isinstance(v, collections.abc.Mapping) and isinstance(dst.get(k), collections.abc.Mapping):
            nested_dict_update(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def load_from_file(path, root=None):
    """YAML -> plain dict with `includes:` resolved (relative to `root` = the AntMMF checkout if the file exists
    there, else relative to the including file) and merged before the file's own keys."""
    if not os.path.isfile(path):
        raise FileNotFoundError(f"No such file: {path}")
    with open(path, "r", encoding="utf-8") as f:
        cfg = yaml.load(os.path.expandvars(f.read()), Loader=yaml.FullLoader) or {}
    merged = {}
    for inc in cfg.get("includes", []) or []:
        cand = os.path.join(root, inc) if root else None
        if cand and os.path.isfile(cand):
            inc_path = cand
        elif os.path.isabs(inc):
            inc_path = inc
        else:
            inc_path = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        nested_dict_update(merged, load_from_file(inc_path, root))
    nested_dict_update(merged, cfg)
    merged.pop("includes", None)
    return merged


class Configuration(collections.abc.MutableMapping):
    def __init__(self, init=None, **kwargs):
        object.__setattr__(self, "_data", {})
        object.__setattr__(self, "_frozen", False)
        if init is not None:
            for k, v in dict(init).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    # ---- construction
    @classmethod
    def from_file(cls, path, with_defaults=False, root=None):
        data = {}
        if with_defaults and os.path.isfile(_DEFAULTS):
            nested_dict_update(data, load_from_file(_DEFAULTS))
        nested_dict_update(data, load_from_file(path, root))
        return cls(data)

    @staticmethod
    def _wrap(v):
        if isinstance(v, Configuration):
            return v
        if isinstance(v, collections.abc.Mapping):
            return Configuration(v)
        if isinstance(v, (list, tuple)):
            return type(v)(Configuration._wrap(x) for x in v)
        return v

    # ---- mapping protocol
    def __getitem__(self, k):
        return self._data[k]

    def __setitem__(self, k, v):
        if self._frozen:
            raise AttributeError(f"Configuration is frozen; cannot set {k}")
        self._data[k] = self._wrap(v)

    def __delitem__(self, k):
        if self._frozen:
            raise AttributeError("Configuration is frozen")
        del self._data[k]

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, k):
        return k in self._data

    def __getattr__(self, k):
        try:
            return self._data[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __repr__(self):
        return f"Configuration({self.to_dict()!r})"

    def __deepcopy__(self, memo):
        return Configuration(copy.deepcopy(self.to_dict(), memo))

    def get(self, k, default=None):
        return self._data.get(k, default)

    def to_dict(self):
        def un(v):
            if isinstance(v, Configuration):
                return {k: un(x) for k, x in v._data.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v

        return un(self)

    # ---- freeze / defrost
    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self._data.values():
            if isinstance(v, Configuration):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self._data.values():
            if isinstance(v, Configuration):
                v.defrost()

    # ---- overrides
    def update_nested(self, other):
        for k, v in dict(other).items():
            if isinstance(v, collections.abc.Mapping) and isinstance(self.get(k), Configuration):
                self[k].update_nested(v)
            else:
                self[k] = v
        return self

    def override_with_cmd_opts(self, opts):
        """`opts` = [dotted.key, value, dotted.key, value, ...]; values go through ast.literal_eval when possible."""
        if not opts:
            return self
        if len(opts) % 2:
            raise ValueError("overrides must come as `key value` pairs")
        for key, val in zip(opts[0::2], opts[1::2]):
            try:
                val = ast.literal_eval(val) if isinstance(val, str) else val
            except (ValueError, SyntaxError):
                pass
            cur = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in cur or not isinstance(cur[p], Configuration):
                    cur[p] = Configuration()
                cur = cur[p]
            cur[parts[-1]] = val
        return self
