"""wittgenstein_amd — MI355X-native engine for Wittgenstein's core.Network scheduler path.

The compute path is libwittgpu.so (hand-written HIP for gfx950, C ABI in include/wittgpu.h); this
package is the thin host-side mirror of the reference's Java surface used by tests and bench.py.
"""
from .core import (Batch, EngineCapacityError, HipError, IllegalArgumentException, IllegalStateException, Network,
                   UnsupportedError)
from .protocols import (GSFSignature, GSFSignatureParameters, Handel, HandelParameters, PingPong,
                        PingPongParameters)

__all__ = ["Network", "Batch", "PingPong", "PingPongParameters", "Handel", "HandelParameters", "GSFSignature",
           "GSFSignatureParameters", "IllegalArgumentException",
           "IllegalStateException", "EngineCapacityError", "HipError", "UnsupportedError"]
