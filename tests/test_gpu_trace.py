"""GPU: layer-by-layer parity of the tensors the fused kernel never writes to HBM -- the input layer's output, the 12 CTFA outputs
(`ctfa_rt` + residual, /root/reference/dnn_model/models/proposed.py:162-196, converter_proposed.py:258-262) and the 6 up-sampling outputs
(`up_sampling`, proposed.py:260-265) -- against the oracle's trace (oracle/nutls_ref.py NutlsRef.trace).  In the fused kernel these rows
go from one op's registers into the LDS image of the next; the state tensors the other parity tests read see them only three ops
downstream.  `nutls_debug_trace` runs the step on the library's PROFILING build of the same kernel source, which copies them out
(include/nutls.h).  One-stream plans of both variants (the packed plans have no profiling build in the default library: their
CTFA / up-sampling / input ops are the same source, checked end to end in test_gpu_packed.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
import nunet_amd
from nunet_amd import NutlsEngine
from nunet_amd import topology as T
from nunet_amd.weights import parse_blob, synthetic_weights, write_blob
from oracle.nutls_ref import NutlsRef

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(1e-12, np.sqrt(np.mean(b ** 2))))


def traced(eng, ref):
    """name -> (kernel's tensor, oracle's tensor) after the step both just took"""
    out = {"input_layer": (eng.debug_get("input_layer", (256, 64)), None)}
    for st in T.ENCODER + T.DECODER:
        out["%s.y" % st.prefix] = (eng.debug_get("%s.y" % st.prefix, (st.f0, 64)), ref.trace["%s.y" % st.prefix].numpy())
    for st in T.DECODER:
        out["%s.up" % st.prefix] = (eng.debug_get("%s.up" % st.prefix, (st.f0, 128)), ref.trace["%s.up" % st.prefix].numpy())
    return out


@pytest.mark.parametrize("variant", ["lstm", "baseline"])
def test_ctfa_upsampling_and_input_layer_outputs_match_the_oracle_trace(variant):
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))["mags_in"]
    B, n = 2, 8
    if variant == "lstm":
        eng, ref = NutlsEngine(batch=B, streams_per_workgroup=1), NutlsRef(batch=B)
    else:
        # (random-init weights stored as the reference's export would store them: conv kernels int8; the oracle gets the de-quantised values)
        blob = write_blob(synthetic_weights("baseline", seed=4321, bias_std=0.1, affine_jitter=0.1), int8_convs=True)
        eng, ref = NutlsEngine(blob, batch=B, variant="baseline"), NutlsRef(parse_blob(blob), batch=B, variant="baseline")
    assert eng.mode == "fused"
    eng.debug_trace(True)
    worst = {}
    for i in range(n):
        x = np.stack([clip[i], clip[(i + 97) % 249]])
        ref.trace = {}
        want_out = ref.step(x).numpy()
        got_out = eng.step(x)
        assert rel_rms(got_out, want_out) < 2e-5, i
        for name, (got, want) in traced(eng, ref).items():
            if want is None:
                continue
            assert got.shape == want.shape, name
            worst[name] = max(worst.get(name, 0.0), rel_rms(got, want))
    assert len(worst) == 18
    bad = {k: v for k, v in worst.items() if not v < 2e-5}
    assert not bad, bad
    # the input layer: 1 -> 64 conv + LN + PReLU of the frame (proposed.py:218-225), restated here from the weights
    wts = {k: v.numpy() for k, v in ref.w.items() if k.startswith("input_layer.")}
    x = np.stack([clip[n - 1], clip[(n - 1 + 97) % 249]]).astype(np.float32)
    y = x[:, :, None] * np.asarray(wts["input_layer.w"]).reshape(1, 1, 64) + np.asarray(wts["input_layer.b"]).reshape(1, 1, 64)
    mu = y.mean(-1, keepdims=True)
    var = ((y - mu) ** 2).mean(-1, keepdims=True)
    y = (y - mu) / np.sqrt(var + 1e-8) * np.asarray(wts["input_layer.gamma"]) + np.asarray(wts["input_layer.beta"])
    a = float(np.asarray(wts["input_layer.alpha"]).reshape(-1)[0])
    y = np.where(y >= 0, y, a * y)
    assert rel_rms(eng.debug_get("input_layer", (256, 64)), y) < 2e-5
    # switching the trace off returns the handle to the production kernel: same results
    eng.debug_trace(False)
    x = np.stack([clip[n], clip[(n + 97) % 249]])
    assert rel_rms(eng.step(x), ref.step(x).numpy()) < 2e-5
    eng.close()


@pytest.mark.gpu
def test_production_timeline_adds_up_and_leaves_fresh_streams():
    """nutls_profile_production: the per-op times of the un-instrumented kernel (launches of the stop twin that end in front of op N,
    differenced) add up to the whole step as the production kernel runs it, a truncated launch is never slower than the whole step, and the
    handle's streams are as new afterwards."""
    import torch
    B = 8
    rng = np.random.default_rng(5)
    x = (0.25 * np.abs(rng.standard_normal((4, B, 256)))).astype(np.float32)
    eng = nunet_amd.NutlsEngine(batch=B, mode="fused")
    for i in range(3):
        eng.step(x[i])                       # (some history for the profile to wipe)
    us = eng.profile_production(reps=2, steps=30)
    n_ops = len(eng.fused_plan())
    assert us.shape == (n_ops + 1,) and np.isfinite(us).all()
    cum = np.cumsum(us)
    assert cum[-1] < 1000.0 and cum[-1] > 100.0            # a step takes 0.25-0.3 ms
    assert (cum[:-1] <= cum[-1] * 1.02).all()
    assert us[1:].sum() > 0.9 * (cum[-1] - us[0])
    # whole step of the stop twin against the production kernel, same handle
    xt = torch.from_numpy(x).cuda(); out = torch.empty(B, 256, device="cuda")
    for i in range(50): eng.step(xt[i % 4], out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(200): eng.step(xt[i % 4], out)
    ev[1].record(); torch.cuda.synchronize()
    step_us = 1e3 * ev[0].elapsed_time(ev[1]) / 200
    assert abs(cum[-1] - step_us) < 0.06 * step_us, (cum[-1], step_us)
    # fresh streams afterwards
    eng.reset()
    fresh = nunet_amd.NutlsEngine(batch=B, mode="fused")
    eng2 = nunet_amd.NutlsEngine(batch=B, mode="fused")
    eng2.step(x[0]); eng2.profile_production(reps=1, steps=4)
    for i in range(3):
        assert np.array_equal(eng2.step(x[i]), fresh.step(x[i])), i
    for e in (eng, eng2, fresh):
        e.close()
    with pytest.raises(ValueError, match="stop twin"):
        b = nunet_amd.NutlsEngine(batch=2, mode="graph")
        try:
            b.profile_production()
        finally:
            b.close()
