"""MI355X-native backend behind tch-geometric's operator surface.

Same layout as the reference package (tch_geometric/__init__.py:1-2): import torch first so that
libtorch is loaded, then re-export the native module.  There is no CPU fallback: the import fails if
the gfx950 library or the host module has not been built (python __graft_entry__.py)."""
import torch  # noqa: F401

from . import _cabi  # noqa: F401  (raises if lib/libtchgeo_hip.so is missing)
from .tch_geometric import *  # noqa: F401,F403
from .tch_geometric import (PanicException, backend_version, graph_cache_clear, graph_cache_info,  # noqa: F401
                            rng_state, seed, set_rng_state)
from .utils import (TEMPORAL_SAMPLE_DYNAMIC, TEMPORAL_SAMPLE_RELATIVE, TEMPORAL_SAMPLE_STATIC,  # noqa: F401
                    TemporalEdgeFilter, UniformEdgeSampler, WeightedEdgeSampler)
