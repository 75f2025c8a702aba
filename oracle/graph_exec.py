"""Oracle A -- float32 op-by-op execution of the reference's shipped graph (TEST INFRASTRUCTURE).

Not part of the product path.  Runs only in the build container, where
``/root/reference/dnn_model/tflite/nutls_lstm.tflite`` exists; it is how the
committed golden vectors under ``tests/golden/`` were produced
(``tests/golden/make_golden.py``) and how oracle B (``oracle/nutls_ref.py``)
is pinned.

What it is: an interpreter for the 26 builtin operator kinds that occur in the
flatbuffer written by ``/root/reference/dnn_model/converter_proposed.py:877-912``
(the traced ``TFL_SIGNITURE.nutls_lstm`` function, ``converter_proposed.py:188-867``).
Weights stored as int8 are de-quantised (``int8 * scale``) and every operator is
evaluated in float32, i.e. what the TF/Keras graph computes with these weights.

PARITY NOTE: the TFLite *runtime* additionally quantises
activations inside its hybrid CONV_2D / FULLY_CONNECTED kernels; that runtime
(third-party, TensorFlow Lite 2.9, not vendored, not installable here) cannot be
run in this image and the reference ships no golden outputs, so bit-level parity
with the TFLite interpreter is *unpinned*.  What *is* pinned is the float
semantics of the shipped graph + weights, independent of any reading of the
model source.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from tools.tflite_reader import TFLiteModel

F32 = np.float32


def _same_pad(in_size: int, k: int, stride: int, dil: int = 1):
    eff = (k - 1) * dil + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + eff - in_size, 0)
    return out, total // 2, total - total // 2


def _conv2d(x, w, b, stride_h, stride_w, padding, dil_h=1, dil_w=1):
    """x NHWC, w OHWI (TFLite layout), float32; padding 0=SAME 1=VALID."""
    n, h, wd, c = x.shape
    o, kh, kw, ci = w.shape
    assert ci == c, (x.shape, w.shape)
    if padding == 0:
        oh, pt, pb = _same_pad(h, kh, stride_h, dil_h)
        ow, pl, pr = _same_pad(wd, kw, stride_w, dil_w)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        oh = (h - ((kh - 1) * dil_h + 1)) // stride_h + 1
        ow = (wd - ((kw - 1) * dil_w + 1)) // stride_w + 1
    out = np.zeros((n, oh, ow, o), dtype=F32)
    for i in range(kh):
        for j in range(kw):
            xs = x[:, i * dil_h: i * dil_h + (oh - 1) * stride_h + 1: stride_h,
                   j * dil_w: j * dil_w + (ow - 1) * stride_w + 1: stride_w, :]
            out += np.matmul(xs.reshape(-1, c), w[:, i, j, :].T).reshape(n, oh, ow, o)
    if b is not None:
        out = out + b
    return out.astype(F32)


def _transpose_conv(out_shape, w, x, b, stride_h, stride_w, padding):
    """TFLite TRANSPOSE_CONV: w OHWI, x NHWC; output cropped to ``out_shape``."""
    n, h, wd, c = x.shape
    o, kh, kw, ci = w.shape
    assert ci == c
    oh, ow = int(out_shape[1]), int(out_shape[2])
    fh, fw = (h - 1) * stride_h + kh, (wd - 1) * stride_w + kw
    full = np.zeros((n, fh, fw, o), dtype=F32)
    for i in range(kh):
        for j in range(kw):
            contrib = np.matmul(x.reshape(-1, c), w[:, i, j, :].T).reshape(n, h, wd, o)
            full[:, i: i + (h - 1) * stride_h + 1: stride_h,
                 j: j + (wd - 1) * stride_w + 1: stride_w, :] += contrib
    if padding == 0:  # SAME: total pad = full - out, split floor/ceil like TFLite ComputePadding
        pt = max(fh - oh, 0) // 2
        pl = max(fw - ow, 0) // 2
    else:
        pt = pl = 0
    out = full[:, pt: pt + oh, pl: pl + ow, :]
    if b is not None:
        out = out + b
    return out.astype(F32)


def _sigmoid(x):
    x = np.asarray(x, dtype=F32)
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e)).astype(F32)


def _fake_quant_rows(x):
    """TF-Lite's per-call activation quantisation of the hybrid kernels (`AsymmetricQuantizeFloats`, one batch row at a time): int8 codes
    q = clamp(round(x / scale) + zp, -128, 127) with scale = (max(x, 0) - min(x, 0)) / 255, returned de-quantised, float32."""
    x = np.asarray(x, dtype=F32)
    flat = x.reshape(x.shape[0], -1).astype(np.float64)
    lo = np.minimum(flat.min(axis=1, keepdims=True), 0.0)
    hi = np.maximum(flat.max(axis=1, keepdims=True), 0.0)
    scale = (hi - lo) / 255.0
    scale[scale == 0.0] = 1.0
    zp = np.clip(np.round(-128.0 - lo / scale), -128, 127)
    q = np.clip(np.round(flat / scale) + zp, -128, 127)
    return ((q - zp) * scale).astype(F32).reshape(x.shape)


def _act(x, code):
    if code == 0:
        return x
    if code == 1:
        return np.maximum(x, 0)
    raise NotImplementedError("fused activation %d" % code)


def _strided_slice(x, begin, end, strides, o):
    idx = []
    nd = x.ndim
    assert o.get("ellipsis_mask", 0) == 0 and o.get("new_axis_mask", 0) == 0
    for d in range(len(begin)):
        st = int(strides[d])
        b = None if (o["begin_mask"] >> d) & 1 else int(begin[d])
        e = None if (o["end_mask"] >> d) & 1 else int(end[d])
        if (o["shrink_axis_mask"] >> d) & 1:
            idx.append(int(begin[d]))
        else:
            idx.append(slice(b, e, st))
    idx += [slice(None)] * (nd - len(begin))
    return x[tuple(idx)]


# ---- third-party arithmetic for the heavy operators (cross-check of the hand-written numpy forms above) ----------
def _conv2d_torch(x, w, b, stride_h, stride_w, padding, dil_h=1, dil_w=1):
    """CONV_2D through torch.nn.functional.conv2d (NHWC / OHWI in and out, explicit TF-style SAME padding)."""
    import torch
    import torch.nn.functional as Fn
    n, h, wd, c = x.shape
    o, kh, kw, ci = w.shape
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=F32)).permute(0, 3, 1, 2)
    if padding == 0:
        _, pt, pb = _same_pad(h, kh, stride_h, dil_h)
        _, pl, pr = _same_pad(wd, kw, stride_w, dil_w)
        xt = Fn.pad(xt, (pl, pr, pt, pb))
    wt = torch.from_numpy(np.ascontiguousarray(w, dtype=F32)).permute(0, 3, 1, 2).contiguous()
    bt = None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=F32))
    y = Fn.conv2d(xt, wt, bt, stride=(stride_h, stride_w), dilation=(dil_h, dil_w))
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def _transpose_conv_torch(out_shape, w, x, b, stride_h, stride_w, padding):
    """TRANSPOSE_CONV through torch.nn.functional.conv_transpose2d (weights OHWI -> torch's [in, out, kh, kw])."""
    import torch
    import torch.nn.functional as Fn
    oh, ow = int(out_shape[1]), int(out_shape[2])
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=F32)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(w, dtype=F32)).permute(3, 0, 1, 2).contiguous()
    full = Fn.conv_transpose2d(xt, wt, None, stride=(stride_h, stride_w))
    fh, fw = full.shape[2], full.shape[3]
    pt = max(fh - oh, 0) // 2 if padding == 0 else 0
    pl = max(fw - ow, 0) // 2 if padding == 0 else 0
    y = full[:, :, pt:pt + oh, pl:pl + ow].permute(0, 2, 3, 1)
    if b is not None:
        y = y + torch.from_numpy(np.ascontiguousarray(b, dtype=F32))
    return y.contiguous().numpy()


class GraphOracle:
    """Callable with the same surface as the reference's TFLite signature runner
    (``/root/reference/dnn_model/interpreter_proposed.py:380, 215-350``).

    ``backend="numpy"``: every operator hand-written in numpy (the generator of the goldens).
    ``backend="torch"``: CONV_2D / TRANSPOSE_CONV / FULLY_CONNECTED / AVERAGE_POOL_2D / LOGISTIC / TANH / PRELU through
    ``torch.nn.functional`` -- third-party arithmetic for everything that carries FLOPs, so the goldens are not pinned to
    hand-written numpy alone (tests/test_oracle.py::test_oracle_a_numpy_ops_agree_with_torch_functional)."""

    def __init__(self, path: str, backend: str = "numpy", hybrid: bool = False):
        """``hybrid``: emulate what the TF-Lite RUNTIME does beyond the float semantics of the graph: its hybrid CONV_2D (v5) and
        FULLY_CONNECTED (v9) kernels quantise the ACTIVATIONS that meet an int8 weight tensor to int8 on every call (per batch row:
        asymmetric, scale = (max - min) / 255 over the row, `tensor_utils::AsymmetricQuantizeFloats`), accumulate in int32 and scale
        back.  Emulated as fake quantisation of those inputs (x -> (q - zp) * scale) in front of the float operator -- the same numbers
        up to fp32 summation order.  This pins nothing (the runtime itself cannot run here); it turns "parity unpinned at the TF-Lite-
        runtime level" into a number: how far ANY float execution of the shipped graph must be from that runtime
        (tests/test_oracle.py::test_hybrid_quantisation_gap_of_the_tflite_runtime)."""
        if backend not in ("numpy", "torch"):
            raise ValueError("backend must be 'numpy' or 'torch'")
        self.backend = backend
        self.hybrid = bool(hybrid)
        self.model = TFLiteModel(path)
        sig = self.model.signatures[0]
        self.key = sig.key
        self.sig_inputs = dict(sig.inputs)
        self.sig_outputs = dict(sig.outputs)
        self.consts: Dict[int, np.ndarray] = {}
        for t in self.model.tensors:
            if t.data is not None:
                self.consts[t.index] = t.data  # raw; DEQUANTIZE / conv dequantise on use
        self._deq: Dict[int, np.ndarray] = {}

    # -----------------------------------------------------------------------------------
    def input_details(self):
        return {n: self.model.tensors[i].shape for n, i in self.sig_inputs.items()}

    def _w(self, idx):
        """float32 view of a (possibly int8) constant."""
        if idx not in self._deq:
            self._deq[idx] = self.model.tensors[idx].dequantized().astype(F32)
        return self._deq[idx]

    def __call__(self, **kwargs) -> Dict[str, np.ndarray]:
        return self.run(kwargs)

    def run(self, feeds: Dict[str, np.ndarray], want: List[int] | None = None, trace=None):
        unknown = set(feeds) - set(self.sig_inputs)
        missing = set(self.sig_inputs) - set(feeds)
        if unknown or missing:
            raise ValueError("bad input names: unknown=%s missing=%s" % (sorted(unknown), sorted(missing)))
        val: Dict[int, np.ndarray] = {}
        for name, idx in self.sig_inputs.items():
            val[idx] = np.asarray(feeds[name], dtype=F32)
        tensors = self.model.tensors

        def get(i):
            if i in val:
                return val[i]
            if i in self.consts:
                t = tensors[i]
                return self._w(i) if t.dtype == np.int8 else self.consts[i]
            raise KeyError("tensor %d (%s) not computed" % (i, tensors[i].name))

        def act_in(op_ins):
            """input activations of a CONV_2D / FULLY_CONNECTED: as they are, or -- hybrid emulation, int8 weight constant -- fake-quantised"""
            x = get(op_ins[0])
            if not self.hybrid or tensors[op_ins[1]].dtype != np.int8:
                return x
            return _fake_quant_rows(x)

        tb = self.backend == "torch"
        if tb:
            import torch
            import torch.nn.functional as Fn
        for op in self.model.ops:
            ins, o, n = op.inputs, op.options, op.name
            if n == "DEQUANTIZE":
                r = self._w(ins[0])
            elif tb and n == "CONV_2D":
                b = get(ins[2]) if len(ins) > 2 and ins[2] >= 0 else None
                r = _act(_conv2d_torch(act_in(ins), get(ins[1]), b, o["stride_h"], o["stride_w"], o["padding"], o["dil_h"], o["dil_w"]), o["act"])
            elif tb and n == "TRANSPOSE_CONV":
                b = get(ins[3]) if len(ins) > 3 and ins[3] >= 0 else None
                r = _transpose_conv_torch(get(ins[0]), get(ins[1]), get(ins[2]), b, o["stride_h"], o["stride_w"], o["padding"])
            elif tb and n == "FULLY_CONNECTED":
                x, w = act_in(ins), get(ins[1])
                b = get(ins[2]) if len(ins) > 2 and ins[2] >= 0 else None
                y = Fn.linear(torch.from_numpy(np.ascontiguousarray(x, dtype=F32)).reshape(-1, w.shape[1]), torch.from_numpy(np.ascontiguousarray(w, dtype=F32)),
                              None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=F32))).numpy()
                if o["keep_num_dims"]:
                    y = y.reshape(x.shape[:-1] + (w.shape[0],))
                r = _act(y.astype(F32), o["act"])
            elif tb and n == "AVERAGE_POOL_2D":
                assert o["padding"] == 1 and o["act"] == 0
                xt = torch.from_numpy(np.ascontiguousarray(get(ins[0]), dtype=F32)).permute(0, 3, 1, 2)
                r = Fn.avg_pool2d(xt, (o["filter_h"], o["filter_w"]), (o["stride_h"], o["stride_w"])).permute(0, 2, 3, 1).contiguous().numpy()
            elif tb and n == "LOGISTIC":
                r = torch.sigmoid(torch.from_numpy(np.ascontiguousarray(get(ins[0]), dtype=F32))).numpy()
            elif tb and n == "TANH":
                r = torch.tanh(torch.from_numpy(np.ascontiguousarray(get(ins[0]), dtype=F32))).numpy()
            elif tb and n == "PRELU":
                x, a = get(ins[0]), get(ins[1])
                r = Fn.prelu(torch.from_numpy(np.ascontiguousarray(x, dtype=F32)), torch.from_numpy(np.ascontiguousarray(a, dtype=F32)).reshape(-1)[:1]).numpy()
            elif n == "CONV_2D":
                b = get(ins[2]) if len(ins) > 2 and ins[2] >= 0 else None
                r = _act(_conv2d(act_in(ins), get(ins[1]), b, o["stride_h"], o["stride_w"],
                                 o["padding"], o["dil_h"], o["dil_w"]), o["act"])
            elif n == "TRANSPOSE_CONV":
                b = get(ins[3]) if len(ins) > 3 and ins[3] >= 0 else None
                r = _transpose_conv(get(ins[0]), get(ins[1]), get(ins[2]), b, o["stride_h"],
                                    o["stride_w"], o["padding"])
            elif n == "FULLY_CONNECTED":
                x, w = act_in(ins), get(ins[1])
                b = get(ins[2]) if len(ins) > 2 and ins[2] >= 0 else None
                y = np.matmul(x.reshape(-1, w.shape[1]), w.T)
                if b is not None:
                    y = y + b
                if o["keep_num_dims"]:
                    y = y.reshape(x.shape[:-1] + (w.shape[0],))
                r = _act(y.astype(F32), o["act"])
            elif n == "MEAN":
                ax = tuple(int(a) for a in np.atleast_1d(get(ins[1])))
                r = np.mean(get(ins[0]), axis=ax, keepdims=bool(o["keep_dims"]), dtype=F32)
            elif n == "REDUCE_PROD":
                ax = tuple(int(a) for a in np.atleast_1d(get(ins[1])))
                r = np.prod(get(ins[0]), axis=ax, keepdims=bool(o["keep_dims"]))
            elif n == "SQUARED_DIFFERENCE":
                d = get(ins[0]) - get(ins[1])
                r = d * d
            elif n == "ADD":
                r = _act(get(ins[0]) + get(ins[1]), o["act"])
            elif n == "SUB":
                r = _act(get(ins[0]) - get(ins[1]), o["act"])
            elif n == "MUL":
                r = _act(get(ins[0]) * get(ins[1]), o["act"])
            elif n == "RSQRT":
                r = (1.0 / np.sqrt(get(ins[0]))).astype(F32)
            elif n == "PRELU":
                x, a = get(ins[0]), get(ins[1])
                r = np.maximum(x, 0) + a * np.minimum(x, 0)
            elif n == "LOGISTIC":
                r = _sigmoid(get(ins[0]))
            elif n == "TANH":
                r = np.tanh(get(ins[0])).astype(F32)
            elif n == "CONCATENATION":
                r = _act(np.concatenate([get(i) for i in ins], axis=o["axis"]), o["act"])
            elif n == "PAD":
                p = get(ins[1])
                r = np.pad(get(ins[0]), [(int(a), int(b)) for a, b in p])
            elif n == "SHAPE":
                r = np.asarray(get(ins[0]).shape, dtype=np.int32)
            elif n == "STRIDED_SLICE":
                r = _strided_slice(get(ins[0]), get(ins[1]), get(ins[2]), get(ins[3]), o)
            elif n == "PACK":
                r = np.stack([get(i) for i in ins], axis=o["axis"])
            elif n == "UNPACK":
                x = get(ins[0])
                parts = [np.take(x, k, axis=o["axis"]) for k in range(o["num"])]
                for k, oi in enumerate(op.outputs):
                    val[oi] = parts[k]
                continue
            elif n == "SPLIT":
                axis = int(get(ins[0]))
                parts = np.split(get(ins[1]), o["num_splits"], axis=axis)
                for k, oi in enumerate(op.outputs):
                    val[oi] = parts[k]
                continue
            elif n == "RESHAPE":
                shp = [int(s) for s in get(ins[1])]
                r = get(ins[0]).reshape(shp)
            elif n == "TRANSPOSE":
                r = np.transpose(get(ins[0]), [int(p) for p in get(ins[1])])
            elif n == "GATHER":
                r = np.take(get(ins[0]), get(ins[1]), axis=o["axis"])
            elif n == "EXPAND_DIMS":
                r = np.expand_dims(get(ins[0]), int(get(ins[1])))
            elif n == "AVERAGE_POOL_2D":
                x = get(ins[0])
                assert o["padding"] == 1 and o["act"] == 0
                fh, fw, sh, sw = o["filter_h"], o["filter_w"], o["stride_h"], o["stride_w"]
                N, H, W, C = x.shape
                oh, ow = (H - fh) // sh + 1, (W - fw) // sw + 1
                r = np.zeros((N, oh, ow, C), dtype=F32)
                for i in range(oh):
                    for j in range(ow):
                        r[:, i, j, :] = x[:, i * sh:i * sh + fh, j * sw:j * sw + fw, :].mean(axis=(1, 2), dtype=F32)
            else:
                raise NotImplementedError(n)
            if r.dtype == np.float64:
                r = r.astype(F32)
            val[op.outputs[0]] = r
            if trace is not None:
                trace(op, r)
        out = {name: val[idx] for name, idx in self.sig_outputs.items()}
        if want:
            out["__extra__"] = {i: val[i] for i in want}
        return out

    # convenience: zero state dict shaped like the reference's ``tflite_out`` seed
    # (interpreter_proposed.py:36-198)
    def zero_feeds(self) -> Dict[str, np.ndarray]:
        feeds = {}
        for name, idx in self.sig_inputs.items():
            shp = tuple(1 if d < 0 else d for d in self.model.tensors[idx].shape)
            feeds[name] = np.zeros(shp, dtype=F32)
        return feeds
