"""Image transforms of the M2 encoder by config key (API of prj/M2_Encoder/vlmo/transforms/__init__.py:19-20:
`keys_to_transforms(cfg["val_transform_keys"], size=cfg["image_size"])`, one transform per key).

Only the resize-to-square family exists here -- it is the step that runs on the MI355X (antmmf.hip.image); the pixelbert variants
of the reference are CPU augmentations that no shipped M2 config selects.
"""
from . import square_transform as _square
from .square_transform import square_transform, square_transform_randaug  # noqa: F401  (re-exported, as in the reference)


def keys_to_transforms(keys: list, size=224):
    made = []
    for key in keys:
        factory = getattr(_square, key, None) if key.startswith("square_transform") else None
        if factory is None:
            raise KeyError(f"unknown transform key {key!r} (available: square_transform, square_transform_randaug)")
        made.append(factory(size=size))
    return made
