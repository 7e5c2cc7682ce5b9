"""lime_amd -- MI355X-native implementation of LiME's ClusterLCP + ClusterBWT_DA hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI + drop-in executables), _lib (ctypes
loader), api (host-side mirror of the reference's two programs), dist (position-range
sharding over GPUs), builder (toy ebwt/lcp/da construction for fixtures).
"""
from .api import Context, cluster_bwt_da, cluster_lcp, reserve, sim_bytes, trim_cache  # noqa: F401
from ._lib import LimeError  # noqa: F401
