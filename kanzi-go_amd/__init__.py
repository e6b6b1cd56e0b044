"""kanzi-go_amd — MI355X-native Kanzi block-compression hot path (host-side Python mirror).

The product is libknz_gpu.so (hand-written gfx950 kernels behind the C ABI of include/knz_gpu.h).
This package is the thin Python mirror of the reference's plugin interfaces for that path
(kanzi.ByteTransform / kanzi.EntropyEncoder / kanzi.EntropyDecoder, v2/Definitions.go:78-179, and
the io.Writer / io.Reader block batch, v2/io/CompressedStream.go:621-710,1614-1744) used by tests
and bench.py. There is NO CPU fallback: if the HIP library is missing, loading fails loudly.
"""
from .api import (  # noqa: F401
    KnzError, Codec, load_library, build_library, library_path, transform_type, entropy_type,
    EntropyEncoder, EntropyDecoder, ByteTransform, BlockBatch,
)
