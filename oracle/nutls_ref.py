"""Oracle B -- batched float32 CPU restatement of the streaming NUNet-TLS-LSTM step
(TEST INFRASTRUCTURE: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this; the product path never does).

Restates, stage by stage, what the reference computes per 16 ms frame:

  blocks   /root/reference/dnn_model/models/proposed.py:162-282
           (ctfa_rt :162, conv_valid :208, inconv :218, spconv_valid :240,
            down_sampling :253, up_sampling :260, reshape_before/after_lstm :268/:277)
  wiring   /root/reference/dnn_model/converter_proposed.py:188-867
           (TFL_SIGNITURE.nutls_lstm: [prev ; cur] time-concat in front of every
            (2,3) conv, LSTM with initial_state, two-level skip connections)

The arithmetic itself lives in a third-party dependency that is absent from
/root/reference and from this image (TensorFlow / TF-Lite 2.9, README.md:79-81),
and the reference ships no golden outputs, so parity with the TFLite *runtime*
(which also int8-quantises activations in its hybrid kernels) is
UNPINNED.  What this file is pinned to instead: oracle A (``oracle/graph_exec.py``),
a float32 op-by-op execution of the reference's own shipped flatbuffer, through
the committed vectors in ``tests/golden/`` (tests/test_oracle.py); agreement is
~1e-6 RMS.

Tensors are channels-last ``[B, F, C]`` torch CPU float32; weights come from the
``.nutlsw`` container (int8 * scale de-quantised -- exactly the DEQUANTIZE the
graph performs).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

import nunet_amd  # noqa: F401  (package alias; topology and weight container are plain data)
from nunet_amd import topology as T
from nunet_amd.weights import load_weights

LN_EPS = 1e-8  # proposed.py:202 (LayerNormalization(epsilon=1e-8))


class NutlsRef:
    """``variant="lstm"``: NUNet-TLS-LSTM (models/proposed.py, trained weights).
    ``variant="baseline"``: NUNet-TLS with the dilated-dense bottleneck
    (models/nunet_tls.py:277-359 blocks, converter_nunet_tls.py:374-411 streaming wiring);
    no trained weights exist for it, pass synthetic ones."""

    def __init__(self, weights: Optional[Dict[str, np.ndarray]] = None, batch: int = 1,
                 ctfa_mode: str = "frame", variant: str = "lstm"):
        self.variant = variant
        w = weights if weights is not None else load_weights()
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}
        self.batch = batch
        self.ctfa_mode = ctfa_mode
        self.state: Dict[str, torch.Tensor] = {}
        self.reset()
        self.trace: Optional[Dict[str, torch.Tensor]] = None

    # ---------------------------------------------------------------- state -------------
    def reset(self):
        """All-zero state, as the reference seeds it (interpreter_proposed.py:36-198)."""
        self.ta_hist = {}
        for base, shp in T.state_specs(self.variant):
            if len(shp) == 1:
                self.state[base] = torch.zeros(self.batch, shp[0])
            elif "_ddb_" in base or base.startswith("ddb_"):
                self.state[base.format("prev")] = torch.zeros(self.batch, *shp)      # [B, d, F, C] oldest first
            else:
                self.state[base.format("prev")] = torch.zeros(self.batch, shp[1], shp[2])

    # ---------------------------------------------------------------- blocks ------------
    def _lnp(self, y, layer):
        """LayerNorm over channels (eps 1e-8) then scalar PReLU (proposed.py:202-203)."""
        g, b, a = self.w[layer + ".gamma"], self.w[layer + ".beta"], self.w[layer + ".alpha"].reshape(())
        mu = y.mean(dim=-1, keepdim=True)
        var = ((y - mu) ** 2).mean(dim=-1, keepdim=True)
        yn = (y - mu) * torch.rsqrt(var + LN_EPS) * g + b
        return torch.clamp(yn, min=0) + a * torch.clamp(yn, max=0)

    def _inconv(self, x, layer):
        """1x1 conv + LN + PReLU (proposed.py:218-225)."""
        w = self.w[layer + ".w"]                       # [64,1,1,Cin]
        y = x @ w.reshape(w.shape[0], -1).t() + self.w[layer + ".b"]
        return self._lnp(y, layer)

    def _conv23(self, prev, cur, layer, stride):
        """Causal (2,3) conv over time taps [prev, cur] and frequency taps f*s-1..f*s+1,
        zero padded by (1,1) in frequency (proposed.py:208-216 / :240-251)."""
        w = self.w[layer + ".w"]                       # [Cout,2,3,Cin] OHWI
        B, F, C = cur.shape
        Fo = F // stride
        y = self.w[layer + ".b"].expand(B, Fo, -1).clone()
        for t, x in enumerate((prev, cur)):
            xp = torch.nn.functional.pad(x, (0, 0, 1, 1))          # pad frequency by 1 each side
            for k in range(3):
                xs = xp[:, k: k + (Fo - 1) * stride + 1: stride, :]
                y = y + xs @ w[:, t, k, :].t()
        return y

    def _el(self, prev, cur, layer):
        return self._lnp(self._conv23(prev, cur, layer, 2), layer)

    def _dl(self, prev, cur, layer):
        """Sub-pixel conv: conv -> Reshape(-1,F,Cin//2,2) -> Permute -> Reshape(-1,2F,out)
        (proposed.py:240-251).  Note the inner dims come from the *input* channel count
        (64 -> (32,2)) even when the conv emits 128 channels."""
        y = self._conv23(prev, cur, layer, 1)          # [B,F,Co]
        B, F, Co = y.shape
        cin = cur.shape[2]
        y = y.reshape(B, -1, F, cin // 2, 2)           # [B,T',F,32,2]; T' = Co/64 (time-like dim)
        y = y.permute(0, 1, 2, 4, 3)                   # [B,T',F,2,32]
        y = y.reshape(B, 2 * F, Co // 2)
        return self._lnp(y, layer)

    def _down(self, x, layer):
        """(1,3) stride-2 conv, TF SAME => pad right only (proposed.py:253-258)."""
        w = self.w[layer + ".w"]                       # [64,1,3,64]
        B, F, C = x.shape
        Fo = F // 2
        xp = torch.nn.functional.pad(x, (0, 0, 0, 1))
        y = self.w[layer + ".b"].expand(B, Fo, -1).clone()
        for k in range(3):
            y = y + xp[:, k: k + (Fo - 1) * 2 + 1: 2, :] @ w[:, 0, k, :].t()
        return y

    def _up(self, x, layer):
        """Conv2DTranspose (1,3) stride 2 SAME: out[2i+k] += W_k x[i], keep 0..2F-1
        (proposed.py:260-265)."""
        w = self.w[layer + ".w"]                       # [128,1,3,128] OHWI
        B, F, C = x.shape
        full = torch.zeros(B, 2 * F + 1, w.shape[0])
        for k in range(3):
            full[:, k: k + 2 * F - 1: 2, :] += x @ w[:, 0, k, :].t()
        return full[:, : 2 * F, :] + self.w[layer + ".b"]

    def _lstm_dense(self, v, lstm, dense, hname, cname):
        """Keras LSTM cell (gates i,f,g,o) + Dense (proposed.py:70-119, converter :234-237)."""
        h, c = self.state[hname], self.state[cname]
        z = v @ self.w[lstm + ".wx"].t() + h @ self.w[lstm + ".wh"].t() + self.w[lstm + ".b"]
        i, f, g, o = z.split(T.LSTM_UNITS, dim=1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        self._new[hname], self._new[cname] = h2, c2
        return h2 @ self.w[dense + ".w"].t() + self.w[dense + ".b"]

    def _bottleneck(self, x, prefix):
        """[B,F,C] -> [B,F,C]: LSTM + Dense (proposed) or the dilated-dense block (baseline)."""
        B, F, C = x.shape
        if self.variant == "lstm":
            names = (prefix + "_lstm", prefix + "_dense", prefix + "_h", prefix + "_c") if prefix else \
                ("lstm", "dense", "state_h", "state_c")
            return self._lstm_dense(x.reshape(B, -1), *names).reshape(B, F, C)
        return self._ddb(x, (prefix + "_ddb") if prefix else "ddb")

    def _prelu(self, y, layer):
        a = self.w[layer + ".alpha"].reshape(())
        return torch.clamp(y, min=0) + a * torch.clamp(y, max=0)

    def _ddb(self, x, tag):
        """Dilated-dense block, streaming form (converter_nunet_tls.py:374-411; blocks
        nunet_tls.py:277-359).  in: (2,3) conv C->G + PReLU.  Block k (dilation d = 2^(k-1) in time
        AND frequency): grouped (groups = G) (2,3) conv over the dense concat [o_{k-1},...,o_0]
        (taps: frame t-d and frame t; bins f-d, f, f+d, zero padded) -> 1x1 conv -> LN -> PReLU.
        out: (2,3) conv G->C + PReLU.  State: one previous frame for in/out, d frames for block k
        (ring shifted by one per step, converter_nunet_tls.py:1420-1427)."""
        B, F, C = x.shape
        G = C // 2
        k_in, k_out = tag + "_prev_in", tag + "_prev_out"
        o = [self._prelu(self._conv23(self.state[k_in][:, 0], x, tag + "_in", 1), tag + "_in")]
        self._new[k_in] = x.unsqueeze(1)
        for k in range(1, T.DDB_BLOCKS + 1):
            d = 1 << (k - 1)
            name = "%s_%d" % (tag, k)
            cur = torch.cat(o[::-1], dim=2)                     # newest first: [o_{k-1}, ..., o_0]  [B,F,k*G]
            hist = self.state["%s_prev%d" % (tag, k)]           # [B,d,F,k*G], index 0 = frame t-d
            wg = self.w[name + ".wg"]                           # [G,2,3,k]
            y = self.w[name + ".bg"].expand(B, F, G).clone()
            for t, src in enumerate((hist[:, 0], cur)):
                xp = torch.nn.functional.pad(src, (0, 0, d, d)).reshape(B, F + 2 * d, G, k)   # group g = channels g*k..g*k+k-1
                for kw in range(3):
                    y = y + (xp[:, kw * d: kw * d + F] * wg[:, t, kw, :]).sum(dim=3)
            self._new["%s_prev%d" % (tag, k)] = torch.cat([hist[:, 1:], cur.unsqueeze(1)], dim=1)
            z = y @ self.w[name + ".w1"].t() + self.w[name + ".b1"]
            g, b = self.w[name + ".gamma"], self.w[name + ".beta"]
            mu = z.mean(dim=-1, keepdim=True)
            var = ((z - mu) ** 2).mean(dim=-1, keepdim=True)
            zn = (z - mu) * torch.rsqrt(var + LN_EPS) * g + b
            o.append(self._prelu(zn, name))
        out = self._prelu(self._conv23(self.state[k_out][:, 0], o[-1], tag + "_out", 1), tag + "_out")
        self._new[k_out] = o[-1].unsqueeze(1)
        return out

    def _mlp_gate(self, m, name):
        w1, w2 = self.w[name + ".w1"], self.w[name + ".w2"]
        hid = torch.relu(m @ w1.reshape(16, 64).t() + self.w[name + ".b1"])
        return torch.sigmoid(hid @ w2.reshape(64, 16).t() + self.w[name + ".b2"])

    def _ctfa(self, x, e0, prefix):
        """ctfa_rt (proposed.py:162-196) + residual (converter_proposed.py:258-262).
        With T = 1 the frequency-attention branch average-pools 31 zero frames and the
        current TA, i.e. sees TA/32 every frame."""
        ta = self._mlp_gate(x.mean(dim=1), prefix + "_ta")          # [B,64]
        if self.ctfa_mode == "causal32":
            # the offline / training model (`ctfa`, proposed.py:125-160): ZeroPadding2D((31,0)) + AveragePooling1D(32,
            # strides=1) over the time-attention vectors of REAL frames = mean of the last 32 TA (zeros before the start)
            hist = self.ta_hist.setdefault(prefix, torch.zeros(self.batch, 31, 64))
            fa = self._mlp_gate((hist.sum(dim=1) + ta) / 32.0, prefix + "_fa")
            self.ta_hist[prefix] = torch.cat([hist[:, 1:], ta.unsqueeze(1)], dim=1)
        elif self.ctfa_mode == "frame":
            fa = self._mlp_gate(ta / 32.0, prefix + "_fa")
        else:
            raise ValueError("ctfa_mode must be 'frame' or 'causal32'")
        return x * (ta * fa).unsqueeze(1) + e0

    # ---------------------------------------------------------------- stage -------------
    def _stage(self, st: T.Stage, x, skips=None):
        tr = self.trace
        e = [self._inconv(x, st.prefix + "_in")]
        for i in range(1, st.depth + 1):
            cur = e[i - 1] if skips is None else torch.cat([e[i - 1], skips[st.depth - i + 1]], dim=2)
            key = "%s_prev%d" % (st.conv_tag, i)
            e.append(self._el(self.state[key], cur, "%s_conv%d" % (st.prefix, i)))
            self._new[key] = cur
        eD = e[st.depth]
        d = self._bottleneck(eD, st.prefix)
        ds = {0: d}
        for j in range(1, st.depth + 1):
            cur = torch.cat([ds[j - 1], e[st.depth - j + 1]], dim=2)
            key = "%s_prev%d" % (st.spconv_tag, j)
            ds[j] = self._dl(self.state[key], cur, "%s_spconv%d" % (st.prefix, j))
            self._new[key] = cur
        y = self._ctfa(ds[st.depth], e[0], st.prefix)
        if tr is not None:
            for i, t in enumerate(e):
                tr["%s.e%d" % (st.prefix, i)] = t
            for j, t in ds.items():
                tr["%s.d%d" % (st.prefix, j)] = t
            tr["%s.y" % st.prefix] = y
        return y, ds

    # ---------------------------------------------------------------- step --------------
    @torch.no_grad()
    def step(self, mag) -> torch.Tensor:
        """mag [B,256] -> enhanced magnitude [B,256]; advances the state by one frame."""
        mag = torch.as_tensor(np.asarray(mag, dtype=np.float32) if not torch.is_tensor(mag) else mag)
        x = self._inconv(mag.reshape(self.batch, T.N_BINS, 1), "input_layer")
        self._new: Dict[str, torch.Tensor] = {}
        enc_d, enc_down = {}, {}
        for st in T.ENCODER:
            y, ds = self._stage(st, x)
            x = self._down(y, st.resample)
            enc_d[st.prefix], enc_down[st.prefix] = ds, x
        u = self._bottleneck(x, "")
        if self.trace is not None:
            self.trace["central.q"] = u
        for st in T.DECODER:
            xin = self._up(torch.cat([u, enc_down[st.pair]], dim=2), st.resample)
            if self.trace is not None:
                self.trace["%s.up" % st.prefix] = xin
            u, _ = self._stage(st, xin, skips=enc_d[st.pair])
        w = self.w["out_conv.w"].reshape(1, 64)
        out = (u @ w.t() + self.w["out_conv.b"]).reshape(self.batch, T.N_BINS)
        self.state.update(self._new)
        return out

    # ---------------------------------------------------------------- compat ------------
    def signature_call(self, **feeds) -> Dict[str, np.ndarray]:
        """Same surface as the reference's signature runner, batch 1
        (interpreter_proposed.py:215-350): named prev/h/c in, named cur/h/c + model_out out."""
        assert self.batch == 1
        names = set(T.input_names())
        if set(feeds) != names:
            raise ValueError("bad input names")
        for base, shp in T.state_specs():
            if len(shp) == 1:
                self.state[base] = torch.from_numpy(np.asarray(feeds[base], np.float32).reshape(1, -1).copy())
            else:
                k = base.format("prev")
                self.state[k] = torch.from_numpy(np.asarray(feeds[k], np.float32).reshape(1, shp[1], shp[2]).copy())
        out = self.step(np.asarray(feeds["input"], np.float32).reshape(1, T.N_BINS))
        res = {"model_out": out.numpy().reshape(1, 1, T.N_BINS, 1)}
        for base, shp in T.state_specs():
            if len(shp) == 1:
                res[base] = self.state[base].numpy().reshape(1, -1)
            else:
                res[base.format("cur")] = self.state[base.format("prev")].numpy().reshape(1, 1, shp[1], shp[2])
        return res
