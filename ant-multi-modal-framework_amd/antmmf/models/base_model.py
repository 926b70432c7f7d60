"""BaseModel contract (reference: antmmf/models/base_model.py:55-220): construct with the model's
`model_attributes.<name>` Configuration, `build()`, `forward(sample_list) -> Mapping` with a "losses" dict;
__call__ checks the mapping and folds registry-configured losses / metrics in (none on the contrastive path:
the model returns its losses itself, as prj/base_vtp's Univl does)."""
import collections.abc
from copy import deepcopy

from torch import nn

from antmmf.common.registry import registry


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.writer = registry.get("writer")
        self.global_config = registry.get("config", {})
        self._is_pretrained = False

    @property
    def is_pretrained(self):
        return self._is_pretrained

    @is_pretrained.setter
    def is_pretrained(self, x):
        self._is_pretrained = x

    def build(self):
        raise NotImplementedError("Build method not implemented in the child model class.")

    def build_for_test(self):
        return self.build()

    def init_losses_and_metrics(self):
        self.losses = None
        self.metrics = None

    @classmethod
    def format_state_key(cls, key):
        return key

    def load_state_dict(self, state_dict, *args, **kwargs):
        copied = deepcopy(state_dict)
        for key in list(copied.keys()):
            copied[self.format_state_key(key)] = copied.pop(key)
        kwargs.pop("strip_head", None)
        return super().load_state_dict(copied, *args, **kwargs)

    def forward(self, sample_list, *args, **kwargs):
        raise NotImplementedError("Forward of the child model class needs to be implemented.")

    def __call__(self, sample_list, *args, **kwargs):
        out = super().__call__(sample_list, *args, **kwargs)
        assert isinstance(out, collections.abc.Mapping), "A dict must be returned from the forward of the model."
        out.setdefault("losses", {})
        out.setdefault("metrics", {})
        return out
