"""nnr -- Python binding of libnnr.so, the MI355X-native NoPe-NeRF render path (see include/nnr.h)."""
from .lib import LAYER_NAMES, LIB_PATH, load  # noqa: F401
from .ops import render_rays  # noqa: F401
