"""build_model (reference: antmmf/models/build.py:9-25)."""
from antmmf.common.registry import registry


def build_model(config, for_test=False):
    name = config.model
    cls = registry.get_model_class(name)
    if cls is None:
        raise ValueError(f"No model registered for name: {name}")
    model = cls(config)
    if hasattr(model, "build"):
        model.build_for_test() if for_test else model.build()
        model.init_losses_and_metrics()
    return model
