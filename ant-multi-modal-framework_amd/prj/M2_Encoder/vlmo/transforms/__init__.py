"""keys_to_transforms (reference: prj/M2_Encoder/vlmo/transforms/__init__.py:10-20)."""
from .square_transform import square_transform, square_transform_randaug

_transforms = {
    "square_transform": square_transform,
    "square_transform_randaug": square_transform_randaug,
}


def keys_to_transforms(keys: list, size=224):
    return [_transforms[key](size=size) for key in keys]
