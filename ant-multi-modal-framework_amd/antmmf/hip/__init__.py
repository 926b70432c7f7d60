"""antmmf.hip -- the MI355X-native operators of the contrastive step (ctypes over libantmmf_hip.so)."""
from . import _lib  # noqa: F401
