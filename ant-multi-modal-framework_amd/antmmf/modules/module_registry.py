"""Per-family nn.Module registries (API of the reference's antmmf/modules/module_registry.py:9-82):

    @VisualEncoder.register()
    class VitImageEncoder(nn.Module): ...
    enc = VisualEncoder(config)        # builds the class named config.type with config.params
    enc.module                         # the built nn.Module (the reference's users take `.module`)
"""
import inspect

from torch import nn


class ModuleRegistry(nn.Module):
    __register_module__ = {}

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if "__register_module__" not in cls.__dict__:
            cls.__register_module__ = {}  # each family keeps its own table

    @classmethod
    def register(cls, module=None):
        if module is None:
            return cls.register
        if not inspect.isclass(module):
            raise ValueError(f"Only class can be registered, but got {module} with type of `{type(module)}`.")
        cls.__register_module__[module.__name__] = module
        return module

    @classmethod
    def get(cls, module_type):
        if module_type not in cls.__register_module__:
            raise ValueError(f"{module_type} is not registered in {cls.__name__}.")
        return cls.__register_module__[module_type]

    def __init__(self, module_type, *args, **kwargs):
        super().__init__()
        self.module = type(self).get(module_type)(*args, **kwargs)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
