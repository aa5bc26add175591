"""simd_dct_amd -- MI355X-native 8x8 block-DCT engine behind the simd_dct.h API.

The product is libmdct_hip.so (hand-written gfx950 HIP kernels + C-ABI, include/mdct.h);
this package is the Python host mirror used by the tests and the benchmark.
"""
from .api import *  # noqa: F401,F403
from .api import QUANTIZE_BASE, MdctError, Prepared, Timer  # noqa: F401
from .sharding import shard_rows, shard_planes, equal_shards  # noqa: F401
