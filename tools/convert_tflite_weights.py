#!/usr/bin/env python3
"""Extract the trained NUNet-TLS weights from a ``.tflite`` the reference exports -- the shipped
``nutls_lstm.tflite`` (LSTM bottlenecks) or a ``nutls.tflite`` (dilated-dense baseline,
``/root/reference/dnn_model/converter_nunet_tls.py:1538-1552``; not shipped, .MISSING_LARGE_BLOBS) --
into this repo's own weight container (``.nutlsw``).

    python tools/convert_tflite_weights.py \
        /root/reference/dnn_model/tflite/nutls_lstm.tflite weights/nutls_lstm.nutlsw

The flatbuffer is the only place the trained parameters exist;
it was written by ``/root/reference/dnn_model/converter_proposed.py:877-912``.
Tensor names inside it follow the Keras layer names; they are
re-keyed here to ``<layer>.<role>``:

    <L>.w      conv kernel, OHWI ``[Cout, kh, kw, Cin]`` (int8 + per-out-channel scale, or f32)
    <L>.b      bias                        <L>.gamma / <L>.beta   LayerNorm scale / offset
    <L>.alpha  PReLU slope (1 scalar)
    <P>_ta.w1/.b1/.w2/.b2, <P>_fa.w1/...   CTFA 64->16->64 MLPs
    <P>_lstm.wx [84,Din] / .wh [84,21] / .b [84]      <P>_dense.w [Dout,21] / .b
    out_conv.w [1,1,1,64] / .b
    baseline only (models/nunet_tls.py:277-359; layer names of its TF-Lite model, :1040-1075 / :1464):
    <T>_in / <T>_out   .w .b .alpha  (2,3) convs + PReLU, no LayerNorm;  T = <P>_ddb or ddb (central)
    <T>_<k>, k = 1..6  .wg [G,2,3,k] / .bg  grouped dilated conv,  .w1 [G,G] / .b1  1x1 conv,  .gamma .beta .alpha

Container layout (little-endian):  magic ``NUTLSW01`` | u32 n | n x { u16 name_len, name,
u8 dtype (0 f32, 1 i8), u8 ndim, u32 dims[ndim], u32 n_scales, f32 scales[n_scales],
payload, zero padding to a 4-byte boundary }.  int8 payloads de-quantise as
``w = q * scale[o]`` (``o`` = index along dim 0, or 0 when ``n_scales == 1``);
all zero-points in the source file are 0 (checked).
"""
from __future__ import annotations

import re
import struct
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.tflite_reader import TFLiteModel  # noqa: E402

MAGIC = b"NUTLSW01"


DDB_BLOCK = re.compile(r"(^|_)ddb_[1-6]$")


class _Reshaped:
    """A constant under another shape (the 1x1 kernel of a dilated-dense block as a [G, G] matrix)."""

    def __init__(self, t, shape):
        self.data = np.asarray(t.data).reshape(shape)
        self.dtype, self.shape = t.dtype, tuple(shape)
        self.scale, self.zero_point, self.qdim = t.scale, t.zero_point, t.qdim


def rekey(model: TFLiteModel) -> "OrderedDict[str, object]":
    out = OrderedDict()
    consts = model.constants()
    # A dilated-dense block is ONE Keras Sequential with TWO convs (nunet_tls.py:286-350): the grouped dilated (2,3) conv
    # [G,2,3,k] and the 1x1 conv [G,1,1,G].  Their biases have the same length; what tells them apart is the `conv2d_N`
    # path component each shares with its kernel.
    pointwise = {}
    for name, t in consts.items():
        parts = name.split("/")
        if DDB_BLOCK.search(parts[0]) and name.endswith("/Conv2D") and len(parts) == 3 and len(t.shape) == 4:
            pointwise[(parts[0], parts[1])] = t.shape[1] == 1 and t.shape[2] == 1
    for name in sorted(consts):
        t = consts[name]
        if t.dtype not in (np.int8, np.float32):
            continue
        parts = name.split("/")
        layer = parts[0]
        key = None
        if "BroadcastTo" in name or name.endswith("/add/y"):
            continue  # all-ones broadcast helpers / LN eps constant
        if DDB_BLOCK.search(layer) and len(parts) >= 3 and (layer, parts[1]) in pointwise:
            pw = pointwise[(layer, parts[1])]
            if name.endswith("/Conv2D"):
                key = layer + (".w1" if pw else ".wg")
                if pw:
                    t = _Reshaped(t, (t.shape[0], t.shape[3]))
            elif name.endswith("BiasAdd/ReadVariableOp"):
                key = layer + (".b1" if pw else ".bg")
        elif re.search(r"/(Conv2D|Conv1D|conv2d_transpose)$", name) and len(parts) in (2, 3):
            if layer.endswith(("_ta", "_fa")):
                key = layer + (".w1" if t.shape[0] == 16 else ".w2")
            else:
                key = layer + ".w"
        elif name.endswith("BiasAdd/ReadVariableOp"):
            if layer.endswith(("_ta", "_fa")):
                key = layer + (".b1" if t.shape[0] == 16 else ".b2")
            else:
                key = layer + ".b"
        elif name.endswith("batchnorm/mul/ReadVariableOp"):
            key = layer + ".gamma"
        elif name.endswith("batchnorm/ReadVariableOp"):
            key = layer + ".beta"
        elif "p_re_lu" in name:
            key = layer + ".alpha"
        elif re.search(r"lstm_cell_\d+/MatMul$", name):
            key = layer + ".wx"
        elif re.search(r"lstm_cell_\d+/MatMul_1\d*$", name):
            key = layer + ".wh"
        elif name.endswith("Tensordot/MatMul"):
            key = layer + ".w"
        if key is None:
            if t.data.size > 1:
                raise ValueError("unmapped weight tensor %s %s" % (name, t.shape))
            continue
        if layer == "conv2d":
            key = "out_conv" + key[len(layer):]
        if key in out:
            raise ValueError("duplicate key %s (from %s)" % (key, name))
        if t.zero_point.size and np.any(t.zero_point != 0):
            raise ValueError("non-zero zero-point in %s" % name)
        out[key] = t
    return out


def write_blob(path: str, tensors) -> int:
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(tensors)))
        for key, t in tensors.items():
            data = np.ascontiguousarray(t.data)
            shape = data.shape if data.ndim else (1,)
            nb = key.encode()
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            is_i8 = data.dtype == np.int8
            f.write(struct.pack("<BB", 1 if is_i8 else 0, len(shape)))
            f.write(struct.pack("<%dI" % len(shape), *shape))
            scales = t.scale.astype(np.float32) if is_i8 else np.zeros((0,), np.float32)
            if is_i8:
                assert scales.size in (1, shape[0]) and (scales.size == 1 or t.qdim == 0), key
            f.write(struct.pack("<I", scales.size))
            f.write(scales.tobytes())
            raw = data.tobytes()
            f.write(raw)
            f.write(b"\0" * ((-len(raw)) % 4))
        return f.tell()


def main(argv):
    src, dst = argv[1], argv[2]
    model = TFLiteModel(src)
    tensors = rekey(model)
    n = write_blob(dst, tensors)
    nparam = sum(int(np.asarray(t.data).size) for t in tensors.values())
    print("wrote %s: %d tensors, %d parameters, %d bytes" % (dst, len(tensors), nparam, n))


if __name__ == "__main__":
    main(sys.argv)
