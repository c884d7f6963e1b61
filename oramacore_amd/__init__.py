"""oramacore_amd — MI355X-native (gfx950) implementation of OramaCore's hybrid-search scoring path.

The product is `csrc/liborama_hip.so` (hand-written HIP kernels behind the C ABI in
include/orama_hip.h).  This Python package is the thin host-side mirror of the reference's
interfaces for that path, used by the tests and bench.py.
"""
from ._native import DTYPE_F16, DTYPE_F32, DTYPE_F32_SHADOW16, METRIC_COSINE, METRIC_L2SQ, OramaError  # noqa: F401
from .context import Context, DeviceBuffer, Stream  # noqa: F401
from .embedding_field import (AllowBitmap, EmbeddingFieldStorage, Model, ResidentAllowBitmap, SearchBatcher,
                              VectorSearchParams)  # noqa: F401
