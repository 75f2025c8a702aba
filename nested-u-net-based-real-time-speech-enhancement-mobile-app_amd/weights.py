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


def quantize_conv_kernels(tensors: Dict[str, np.ndarray]) -> Dict[str, object]:
    """What ``tf.lite.Optimize.DEFAULT`` without a representative dataset (dynamic-range quantisation, the reference's
    export: converter_proposed.py:901, converter_nunet_tls.py:1552) does to the encoder / decoder Conv2D kernels:
    symmetric int8 per OUTPUT channel, ``scale = max|w| / 127``, ``q = round(w / scale)``; tensors below 1024
    elements stay float, as in TF-Lite.  The dilated-dense blocks' kernels are left float (the grouped convs are
    tiny and the block runs in fp32 on every path).  Returns name -> ndarray or ``(int8 array, scales)``."""
    out: Dict[str, object] = {}
    for name, arr in tensors.items():
        a = np.asarray(arr)
        if name.endswith(".w") and a.ndim == 4 and a.size >= 1024 and "ddb" not in name:
            amax = np.abs(a).reshape(a.shape[0], -1).max(axis=1)
            scale = np.where(amax > 0, amax / 127.0, 1.0).astype(np.float32)
            q = np.clip(np.rint(a / scale.reshape(-1, 1, 1, 1)), -127, 127).astype(np.int8)
            out[name] = (q, scale)
        else:
            out[name] = arr
    return out


def write_blob(tensors: Dict[str, object], int8_convs: bool = False) -> bytes:
    """Serialise tensors into a NUTLSW01 container: float32 arrays as they are, ``(int8 array, scales)`` tuples as int8
    payload + scales.  ``int8_convs``: quantise the conv kernels first (`quantize_conv_kernels`) -- the form the fused
    kernel takes (conv kernels stay int8 on the device)."""
    if int8_convs:
        tensors = quantize_conv_kernels(tensors)
    out = [MAGIC, struct.pack("<I", len(tensors))]
    for name, arr in tensors.items():
        nb = name.encode()
        if isinstance(arr, tuple):
            q = np.ascontiguousarray(arr[0], dtype=np.int8)
            sc = np.ascontiguousarray(arr[1], dtype=np.float32).reshape(-1)
            out += [struct.pack("<H", len(nb)), nb, struct.pack("<BB", 1, q.ndim), struct.pack("<%dI" % q.ndim, *q.shape),
                    struct.pack("<I", sc.size), sc.tobytes(), q.tobytes(), b"\0" * ((-q.size) % 4)]
            continue
        a = np.ascontiguousarray(arr, dtype=np.float32)
        shape = a.shape if a.ndim else (1,)
        out += [struct.pack("<H", len(nb)), nb, struct.pack("<BB", 0, len(shape)),
                struct.pack("<%dI" % len(shape), *shape), struct.pack("<I", 0), a.tobytes()]
    return b"".join(out)


def synthetic_weights(variant: str = "baseline", seed: int = 4321, bias_std: float = 0.0,
                      affine_jitter: float = 0.0) -> Dict[str, np.ndarray]:
    """Random-init weights of the NUNet-TLS architecture (BASELINE config 3: the dilated-dense
    baseline has no trained weights anywhere).  Conv kernels N(0, 1/fan_in), biases
    0, LayerNorm gamma 1 / beta 0, PReLU alpha 0.25 (the Keras initialisers of
    models/nunet_tls.py:32); ``bias_std`` / ``affine_jitter`` > 0 randomise biases and the LN /
    PReLU parameters so parity tests exercise those paths too."""
    from . import topology as T
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def conv(name, cout, kh, kw, cin, ln=None, prelu=False, fan=None):
        fan_in = fan if fan else kh * kw * cin
        w[name + ".w"] = (rng.standard_normal((cout, kh, kw, cin)) / np.sqrt(fan_in)).astype(np.float32)
        w[name + ".b"] = (bias_std * rng.standard_normal(cout)).astype(np.float32)
        if ln:
            w[name + ".gamma"] = (1.0 + affine_jitter * rng.standard_normal(ln)).astype(np.float32)
            w[name + ".beta"] = (affine_jitter * rng.standard_normal(ln)).astype(np.float32)
        if ln or prelu:
            w[name + ".alpha"] = np.float32([0.25 + affine_jitter * rng.standard_normal()]).reshape(1, 1, 1)

    def gate(name):
        w[name + ".w1"] = (rng.standard_normal((16, 1, 1, 64)) / 8.0).astype(np.float32)
        w[name + ".b1"] = (bias_std * rng.standard_normal(16)).astype(np.float32)
        w[name + ".w2"] = (rng.standard_normal((64, 1, 1, 16)) / 4.0).astype(np.float32)
        w[name + ".b2"] = (bias_std * rng.standard_normal(64)).astype(np.float32)

    def bottleneck(prefix, f, c):
        if variant == "lstm":
            din = f * c
            lp, dp = (prefix + "_lstm", prefix + "_dense") if prefix else ("lstm", "dense")
            w[lp + ".wx"] = (rng.standard_normal((84, din)) / np.sqrt(din)).astype(np.float32)
            w[lp + ".wh"] = (rng.standard_normal((84, 21)) / np.sqrt(21)).astype(np.float32)
            w[lp + ".b"] = (bias_std * rng.standard_normal(84)).astype(np.float32)
            w[dp + ".w"] = (rng.standard_normal((din, 21)) / np.sqrt(21)).astype(np.float32)
            w[dp + ".b"] = (bias_std * rng.standard_normal(din)).astype(np.float32)
            return
        g = c // 2
        tag = (prefix + "_ddb") if prefix else "ddb"
        conv(tag + "_in", g, 2, 3, c, prelu=True)
        for k in range(1, T.DDB_BLOCKS + 1):
            n = "%s_%d" % (tag, k)
            # grouped (groups = g) dilated (2,3) conv: filter j sees k channels -> [g, 2, 3, k]
            w[n + ".wg"] = (rng.standard_normal((g, 2, 3, k)) / np.sqrt(6 * k)).astype(np.float32)
            w[n + ".bg"] = (bias_std * rng.standard_normal(g)).astype(np.float32)
            w[n + ".w1"] = (rng.standard_normal((g, g)) / np.sqrt(g)).astype(np.float32)
            w[n + ".b1"] = (bias_std * rng.standard_normal(g)).astype(np.float32)
            w[n + ".gamma"] = (1.0 + affine_jitter * rng.standard_normal(g)).astype(np.float32)
            w[n + ".beta"] = (affine_jitter * rng.standard_normal(g)).astype(np.float32)
            w[n + ".alpha"] = np.float32([0.25 + affine_jitter * rng.standard_normal()]).reshape(1, 1, 1)
        conv(tag + "_out", c, 2, 3, g, prelu=True)

    conv("input_layer", 64, 1, 1, 1, ln=64)
    for st in T.STAGES:
        P = st.prefix
        conv(P + "_in", 64, 1, 1, 128 if st.is_decoder else 64, ln=64)
        for i in range(1, st.depth + 1):
            conv("%s_conv%d" % (P, i), 32, 2, 3, T.conv_state_shape(st, i)[1], ln=32)
        for j in range(1, st.depth + 1):
            co = 64 if j < st.depth else 128
            conv("%s_spconv%d" % (P, j), co, 2, 3, 64, ln=co // 2)
        gate(P + "_ta"); gate(P + "_fa")
        if st.is_decoder:
            conv(st.resample, 128, 1, 3, 128)
        else:
            conv(st.resample, 64, 1, 3, 64)
    bn = T.bottlenecks()
    for prefix, f, c in bn:
        bottleneck(prefix, f, c)
    w["out_conv.w"] = (rng.standard_normal((1, 1, 1, 64)) / 8.0).astype(np.float32)
    w["out_conv.b"] = (bias_std * rng.standard_normal(1)).astype(np.float32)
    return w
