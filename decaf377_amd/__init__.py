"""decaf377_amd -- MI355X-native batch group-operation engine for decaf377.

Host-side mirror of the reference crate's public hot-path API (penumbra-zone/decaf377:
`Encoding`, `Element`, `Fq`, `Fr`, `EncodingError`) as batch operations over packed
32-byte records, backed by hand-written gfx950 kernels behind a C ABI
(include/decaf377_amd.h).  See DESIGN.md."""
from ._native import LIB_PATH, NativeError, StarvedError, load  # noqa: F401
from .engine import (  # noqa: F401
    Context,
    Element,
    Encoding,
    EncodingError,
    Fq,
    Fr,
    default_context,
)

__all__ = ["Context", "Element", "Encoding", "EncodingError", "Fq", "Fr", "default_context",
           "NativeError", "StarvedError", "load", "LIB_PATH"]
