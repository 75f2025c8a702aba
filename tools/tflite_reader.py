"""Minimal FlatBuffer reader for a TFLite model file (weight-file tooling).

Used by ``tools/convert_tflite_weights.py`` (weight extraction) and by
``oracle/graph_exec.py`` (oracle A); never imported by the product path, which
only consumes the ``.nutlsw`` weight blob.  It exists because the only trained weights of the reference's
hot path live inside ``/root/reference/dnn_model/tflite/nutls_lstm.tflite``
(produced by ``/root/reference/dnn_model/converter_proposed.py:877-912``) and
neither ``tensorflow`` nor ``flatbuffers`` is installed in this image.

It decodes just the tables the NUNet-TLS-LSTM graph uses (schema: TFLite
``schema.fbs`` v3 -- third-party, un-vendored; field ids as listed in
SURVEY.md Appendix E).  Only ``struct`` + ``numpy`` are needed.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# builtin operator codes that occur in the shipped graph (schema.fbs BuiltinOperator)
BUILTIN = {
    0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D", 6: "DEQUANTIZE",
    9: "FULLY_CONNECTED", 14: "LOGISTIC", 18: "MUL", 22: "RESHAPE", 28: "TANH", 34: "PAD",
    36: "GATHER", 39: "TRANSPOSE", 40: "MEAN", 41: "SUB", 45: "STRIDED_SLICE", 49: "SPLIT",
    54: "PRELU", 67: "TRANSPOSE_CONV", 70: "EXPAND_DIMS", 76: "RSQRT", 77: "SHAPE",
    81: "REDUCE_PROD", 83: "PACK", 88: "UNPACK", 99: "SQUARED_DIFFERENCE",
}

_TENSOR_DTYPES = {0: np.float32, 2: np.int32, 3: np.uint8, 4: np.int64, 9: np.int8}


class _FB:
    """Just enough of the FlatBuffer wire format (little-endian, vtables)."""

    def __init__(self, buf: bytes):
        self.b = buf

    def u8(self, o): return self.b[o]
    def i8(self, o): return struct.unpack_from("<b", self.b, o)[0]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def i32(self, o): return struct.unpack_from("<i", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]

    def root(self) -> int:
        return self.u32(0)

    def field(self, table: int, idx: int) -> int:
        """Absolute offset of field ``idx`` inside ``table`` or 0 when absent."""
        vt = table - self.i32(table)
        vsize = self.u16(vt)
        slot = 4 + 2 * idx
        if slot >= vsize:
            return 0
        off = self.u16(vt + slot)
        return table + off if off else 0

    def indirect(self, o: int) -> int:
        return o + self.u32(o)

    def scalar(self, table, idx, kind, default=0):
        o = self.field(table, idx)
        if not o:
            return default
        return getattr(self, kind)(o)

    def table(self, table, idx) -> int:
        o = self.field(table, idx)
        return self.indirect(o) if o else 0

    def vector(self, table, idx):
        """(start offset of element 0, length) or (0, 0)."""
        o = self.field(table, idx)
        if not o:
            return 0, 0
        v = self.indirect(o)
        return v + 4, self.u32(v)

    def string(self, table, idx) -> str:
        s, n = self.vector(table, idx)
        return self.b[s:s + n].decode("utf-8") if s else ""

    def vec_np(self, table, idx, dtype) -> np.ndarray:
        s, n = self.vector(table, idx)
        if not s:
            return np.zeros((0,), dtype=dtype)
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=s).copy()

    def vec_tables(self, table, idx) -> List[int]:
        s, n = self.vector(table, idx)
        return [self.indirect(s + 4 * i) for i in range(n)]


@dataclass
class TensorInfo:
    index: int
    name: str
    shape: tuple
    dtype: type
    buffer: int
    scale: np.ndarray
    zero_point: np.ndarray
    qdim: int
    data: Optional[np.ndarray] = None  # constant payload (raw dtype), None for activations

    def dequantized(self) -> np.ndarray:
        """fp32 view of a constant: int8 * scale (zero-points are all 0 in this file)."""
        assert self.data is not None
        if self.data.dtype != np.int8:
            return self.data
        w = self.data.astype(np.float32)
        if self.scale.size == 1:
            return w * self.scale[0]
        shp = [1] * w.ndim
        shp[self.qdim] = -1
        return w * self.scale.reshape(shp)


@dataclass
class OpInfo:
    index: int
    code: int
    name: str
    inputs: List[int]
    outputs: List[int]
    options: Dict[str, int] = field(default_factory=dict)


@dataclass
class Signature:
    key: str
    inputs: Dict[str, int]
    outputs: Dict[str, int]


class TFLiteModel:
    """Parsed view of one ``.tflite`` file: tensors, operators, signature."""

    def __init__(self, path: str):
        with open(path, "rb") as f:
            raw = f.read()
        if raw[4:8] != b"TFL3":
            raise ValueError("not a TFLite flatbuffer (missing TFL3 identifier)")
        fb = self.fb = _FB(raw)
        model = fb.root()
        self.version = fb.scalar(model, 0, "u32")
        # operator codes
        self.opcodes = []
        for t in fb.vec_tables(model, 1):
            dep = fb.scalar(t, 0, "i8")
            new = fb.scalar(t, 3, "i32")
            self.opcodes.append((max(dep, new), fb.scalar(t, 2, "i32", 1)))
        # buffers
        buffers = []
        for t in fb.vec_tables(model, 4):
            s, n = fb.vector(t, 0)
            buffers.append((s, n))
        sub = fb.vec_tables(model, 2)
        if len(sub) != 1:
            raise ValueError("expected exactly one subgraph")
        sg = sub[0]
        self.tensors: List[TensorInfo] = []
        for i, t in enumerate(fb.vec_tables(sg, 0)):
            shape = tuple(int(x) for x in fb.vec_np(t, 0, np.int32))
            tcode = fb.scalar(t, 1, "i8")
            dtype = _TENSOR_DTYPES[tcode]
            bidx = fb.scalar(t, 2, "u32")
            name = fb.string(t, 3)
            q = fb.table(t, 4)
            scale = fb.vec_np(q, 2, np.float32) if q else np.zeros((0,), np.float32)
            zp = fb.vec_np(q, 3, np.int64) if q else np.zeros((0,), np.int64)
            qdim = fb.scalar(q, 6, "i32") if q else 0
            data = None
            s, n = buffers[bidx]
            if n:
                cnt = n // np.dtype(dtype).itemsize
                data = np.frombuffer(raw, dtype=dtype, count=cnt, offset=s).copy()
                data = data.reshape(shape) if shape else data.reshape(())
            self.tensors.append(TensorInfo(i, name, shape, dtype, bidx, scale, zp, qdim, data))
        self.inputs = [int(x) for x in fb.vec_np(sg, 1, np.int32)]
        self.outputs = [int(x) for x in fb.vec_np(sg, 2, np.int32)]
        self.ops: List[OpInfo] = []
        for i, t in enumerate(fb.vec_tables(sg, 3)):
            code, _ver = self.opcodes[fb.scalar(t, 0, "u32")]
            ins = [int(x) for x in fb.vec_np(t, 1, np.int32)]
            outs = [int(x) for x in fb.vec_np(t, 2, np.int32)]
            name = BUILTIN.get(code, "OP_%d" % code)
            self.ops.append(OpInfo(i, code, name, ins, outs, self._options(name, fb.table(t, 4))))
        self.signatures: List[Signature] = []
        for t in fb.vec_tables(model, 7):
            def tmap(idx):
                return {fb.string(m, 0): fb.scalar(m, 1, "u32") for m in fb.vec_tables(t, idx)}
            self.signatures.append(Signature(fb.string(t, 2), tmap(0), tmap(1)))

    def _options(self, name: str, o: int) -> Dict[str, int]:
        fb = self.fb
        if not o:
            return {}
        g = lambda i, kind="i32", d=0: fb.scalar(o, i, kind, d)  # noqa: E731
        if name == "CONV_2D":
            return dict(padding=g(0, "i8"), stride_w=g(1), stride_h=g(2), act=g(3, "i8"),
                        dil_w=g(4, "i32", 1), dil_h=g(5, "i32", 1))
        if name == "TRANSPOSE_CONV":
            return dict(padding=g(0, "i8"), stride_w=g(1), stride_h=g(2))
        if name == "FULLY_CONNECTED":
            return dict(act=g(0, "i8"), keep_num_dims=g(2, "u8"), asym=g(3, "u8"))
        if name == "AVERAGE_POOL_2D":
            return dict(padding=g(0, "i8"), stride_w=g(1), stride_h=g(2), filter_w=g(3),
                        filter_h=g(4), act=g(5, "i8"))
        if name == "CONCATENATION":
            return dict(axis=g(0), act=g(1, "i8"))
        if name in ("MEAN", "REDUCE_PROD"):
            return dict(keep_dims=g(0, "u8"))
        if name == "PACK":
            return dict(values_count=g(0), axis=g(1))
        if name == "UNPACK":
            return dict(num=g(0), axis=g(1))
        if name == "SPLIT":
            return dict(num_splits=g(0))
        if name == "STRIDED_SLICE":
            return dict(begin_mask=g(0), end_mask=g(1), ellipsis_mask=g(2), new_axis_mask=g(3),
                        shrink_axis_mask=g(4))
        if name == "GATHER":
            return dict(axis=g(0), batch_dims=g(1))
        if name in ("ADD", "MUL", "SUB"):
            return dict(act=g(0, "i8"))
        return {}

    # convenience ---------------------------------------------------------------
    def constants(self) -> Dict[str, TensorInfo]:
        return {t.name: t for t in self.tensors if t.data is not None}

    def parameter_count(self) -> int:
        """Number of trained scalars (weights held as int8 or fp32 constants with >1 dims
        or belonging to a layer); helper constants (shapes, axes, eps) are int32/scalars."""
        n = 0
        for t in self.tensors:
            if t.data is None or t.dtype not in (np.int8, np.float32):
                continue
            n += int(t.data.size)
        return n
