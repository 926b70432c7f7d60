"""build_trainer (reference: antmmf/trainers/build.py:12-26)."""
from antmmf.common.registry import registry
from antmmf.trainers import base_trainer, retrieval_trainer  # noqa: F401  (register "base_trainer" / "retrieval_trainer")


def build_trainer(config, *args, **kwargs):
    name = config.training_parameters.trainer
    cls = registry.get_trainer_class(name)
    if cls is None:
        raise ValueError(f"No trainer registered for name: {name}")
    return cls(config, *args, **kwargs)
