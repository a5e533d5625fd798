"""MI355X-native backend behind tch-geometric's operator surface.

Mirrors the reference package layout (tch_geometric/__init__.py:1-2 re-exports the native module)."""
import torch  # noqa: F401  (the reference imports torch first so libtorch symbols are loaded)

from . import _cabi  # noqa: F401  raises if the gfx950 library is missing -- no CPU fallback
