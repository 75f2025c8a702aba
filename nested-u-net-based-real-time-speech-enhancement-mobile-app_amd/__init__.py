"""MI355X-native NUNet-TLS frame-by-frame forward pass (hot path only).

Host side mirrors the reference's ``interpreter_*`` call surface
(``/root/reference/dnn_model/interpreter_proposed.py:215-350, 380``); all
arithmetic runs in the HIP library built from ``csrc/`` behind the C ABI of
``include/nutls.h``.  There is no CPU fallback: importing works anywhere, but
creating an engine without the built library or without a GPU raises.
"""
from . import topology, weights, stream_enhance  # noqa: F401
from .runner import NutlsEngine, NutlsOffline, NutlsRunner, host_alloc, load_library  # noqa: F401

__all__ = ["topology", "weights", "stream_enhance", "NutlsEngine", "NutlsRunner", "NutlsOffline", "load_library", "host_alloc"]
