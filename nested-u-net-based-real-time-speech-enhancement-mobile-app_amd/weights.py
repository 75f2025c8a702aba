"""Reader for the ``.nutlsw`` weight container (see ``tools/convert_tflite_weights.py``
for the layout).  Host-side plumbing only: the HIP library parses the same bytes
itself (``csrc/weights.cpp``); this reader exists so Python callers can inspect
the tensors and so the test oracle can be fed the identical parameters.
"""
from __future__ import annotations

import os
import struct
from typing import Dict

import numpy as np

MAGIC = b"NUTLSW01"
DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                               "weights", "nutls_lstm.nutlsw")


def read_blob(path_or_bytes) -> bytes:
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        return bytes(path_or_bytes)
    with open(path_or_bytes, "rb") as f:
        return f.read()


def parse_blob(blob: bytes, dequantize: bool = True) -> Dict[str, np.ndarray]:
    """name -> ndarray.  With ``dequantize`` int8 tensors come back as float32
    ``q * scale`` (the only interpretation the model uses); otherwise as
    ``(int8 array, scales)`` tuples."""
    if blob[:8] != MAGIC:
        raise ValueError("not a NUTLSW01 weight container")
    (n,) = struct.unpack_from("<I", blob, 8)
    off = 12
    out: Dict[str, np.ndarray] = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, off); off += 2
        name = blob[off:off + ln].decode(); off += ln
        dtype, ndim = struct.unpack_from("<BB", blob, off); off += 2
        dims = struct.unpack_from("<%dI" % ndim, blob, off); off += 4 * ndim
        (ns,) = struct.unpack_from("<I", blob, off); off += 4
        scales = np.frombuffer(blob, np.float32, ns, off).copy(); off += 4 * ns
        cnt = int(np.prod(dims))
        if dtype == 0:
            arr = np.frombuffer(blob, np.float32, cnt, off).reshape(dims).copy(); off += 4 * cnt
        elif dtype == 1:
            q = np.frombuffer(blob, np.int8, cnt, off).reshape(dims).copy(); off += cnt
            off += (-cnt) % 4
            if dequantize:
                if ns == 1:
                    arr = q.astype(np.float32) * scales[0]
                else:
                    arr = q.astype(np.float32) * scales.reshape((-1,) + (1,) * (ndim - 1))
            else:
                arr = (q, scales)
        else:
            raise ValueError("bad dtype code %d for %s" % (dtype, name))
        out[name] = arr
    if off != len(blob):
        raise ValueError("trailing bytes in weight container")
    return out


def load_weights(path: str = DEFAULT_WEIGHTS) -> Dict[str, np.ndarray]:
    return parse_blob(read_blob(path))
