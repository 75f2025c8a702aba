"""``-m gpu``: offline / block mode through the C ABI: T frames of one utterance per call must
equal the frame-by-frame streaming result (same function, models/proposed.py offline semantics with the streaming
CTFA) -- against the golden clip (oracle A outputs), the streaming HIP engine, and across block boundaries."""
import os

import numpy as np
import pytest

from nunet_amd import NutlsEngine, NutlsOffline

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.fixture(scope="module")
def clip():
    return np.load(os.path.join(GOLDEN, "clip_4s.npz"))


@pytest.mark.parametrize("max_frames", [256, 64, 7])
def test_block_mode_equals_golden_and_streaming(clip, max_frames):
    """249 frames in blocks of 256 (one call), 64 (carry across 4 calls) and 7 (ragged last block)."""
    off = NutlsOffline(max_frames=max_frames)
    got = off.process(clip["mags_in"])
    off.close()
    assert rms(got, clip["mags_out"]) < 2e-5
    eng = NutlsEngine(batch=1)
    stream = np.stack([eng.step(clip["mags_in"][i:i + 1])[0] for i in range(40)])
    eng.close()
    assert rms(got[:40], stream) < 1e-6


def test_ragged_block_lengths_cover_every_tail_of_the_scan(clip):
    """The LSTM scan walks 16 frames per round (two prefetch groups of eight) and finishes with guarded steps: block
    lengths below one group, between one and two groups, one past a round ... on ONE utterance, state carried."""
    sizes = [1, 9, 15, 16, 17, 23, 8, 31, 33, 40, 2, 14]
    assert sum(sizes) <= len(clip["mags_in"])
    off = NutlsOffline(max_frames=40)
    out, t = [], 0
    for n in sizes:
        out.append(off.process(clip["mags_in"][t:t + n]))
        assert out[-1].shape == (n, 256)
        t += n
    off.close()
    got = np.concatenate(out)
    assert rms(got, clip["mags_out"][:t]) < 2e-5
    whole = NutlsOffline(max_frames=256)
    want = whole.process(clip["mags_in"][:t])
    whole.close()
    assert rms(got, want) < 1e-6


def test_block_mode_bf16_pipe_equals_the_fp32_mfma_kernels(clip, monkeypatch):
    """The block mode's convs run on the bf16 matrix pipe (conv_bf16x3_kernel: int8 weights widened to bf16, activations
    split error-free into three bf16 pieces -- every product exact, fp32 accumulation); NUTLS_OFFLINE_FP32=1 keeps the
    fp32-MFMA kernels.  Same function, summation order apart."""
    off = NutlsOffline(max_frames=128)
    got = off.process(clip["mags_in"][:128])
    off.close()
    monkeypatch.setenv("NUTLS_OFFLINE_FP32", "1")
    ref = NutlsOffline(max_frames=128)
    want = ref.process(clip["mags_in"][:128])
    ref.close()
    assert rms(got, want) < 1e-6
    assert rms(got, clip["mags_out"][:128]) < 2e-5 and rms(want, clip["mags_out"][:128]) < 2e-5
    assert not np.array_equal(got, want)        # (two different kernels did run)


def test_state_carries_between_calls_and_reset_restarts(clip):
    off = NutlsOffline(max_frames=32)
    a = off.process(clip["mags_in"][:32])
    b = off.process(clip["mags_in"][32:64])            # continues the utterance
    assert rms(np.concatenate([a, b]), clip["mags_out"][:64]) < 2e-5
    off.reset()
    again = off.process(clip["mags_in"][:32])
    np.testing.assert_array_equal(a, again)
    with pytest.raises(ValueError):
        off.process(np.zeros((4, 255), np.float32))
    off.close()


def test_streaming_entry_points_reject_offline_handles_and_vice_versa(clip):
    import ctypes
    from nunet_amd.runner import _fptr
    off = NutlsOffline(max_frames=8)
    x = np.zeros((9, 256), np.float32)
    assert off._lib.nutls_step_host(off._h, _fptr(x), _fptr(x.copy())) != 0
    assert off._lib.nutls_process_block_host(off._h, _fptr(x), _fptr(x.copy()), 9) != 0      # > max_frames
    off.close()
    eng = NutlsEngine(batch=2)
    y = np.zeros((2, 256), np.float32)
    assert eng._lib.nutls_process_block_host(eng._h, _fptr(y), _fptr(y.copy()), 2) != 0
    eng.close()


def test_offline_handle_state_get_set_addresses_the_carried_state():
    """state_get / state_set on an offline handle read / write the state carried from block to block (one utterance)."""
    import numpy as np
    import nunet_amd
    from conftest import GOLDEN
    import os
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    off = nunet_amd.NutlsOffline(max_frames=16)
    eng = nunet_amd.NutlsEngine(batch=1)
    off.process(clip["mags_in"][:16])
    for i in range(16):
        eng.step(clip["mags_in"][i:i + 1])
    lib, h = off._lib, off._h
    import ctypes
    for name, n in (("state_h", 21), ("msfe6_ee_prev1", 256 * 64), ("msfe3_dd_prev2", 2 * 64)):
        a = np.zeros(n, np.float32)
        assert lib.nutls_state_get(h, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n) == 0, lib.nutls_last_error()
        # (two different kernels, two summation orders of the same fp32 sums -- the block mode's small layers split K over four
        #  waves: a LayerNorm output near zero differs by up to ~1e-5 absolute)
        np.testing.assert_allclose(a, eng.state_get(name).reshape(-1), rtol=1e-4, atol=3e-5, err_msg=name)
    # zero one state on both and continue: still the same function
    z = np.zeros(21, np.float32)
    assert lib.nutls_state_set(h, b"state_h", z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 21) == 0
    eng.state_set("state_h", z.reshape(1, 21))
    o1 = off.process(clip["mags_in"][16:24])
    o2 = np.concatenate([eng.step(clip["mags_in"][i:i + 1]) for i in range(16, 24)])
    assert float(np.sqrt(np.mean((o1 - o2) ** 2))) < 2e-5
    off.close(); eng.close()


def test_offline_causal32_ctfa_matches_oracle_across_block_boundaries(clip):
    """The non-default CTFA option of the offline mode (true 32-frame causal average of the time attention,
    models/proposed.py:143-147) against oracle B in the same mode, in blocks that do and do not divide 31."""
    from oracle.nutls_ref import NutlsRef
    n = 80
    ref = NutlsRef(batch=1, ctfa_mode="causal32")
    want = np.concatenate([ref.step(clip["mags_in"][i:i + 1]).numpy() for i in range(n)])
    for T_ in (32, 7):
        off = NutlsOffline(max_frames=T_, ctfa_mode="causal32")
        got = off.process(clip["mags_in"][:n])
        assert rms(got, want) < 2e-5, T_
        off.reset()                                   # a reset clears the history: same result again
        assert rms(off.process(clip["mags_in"][:n]), want) < 2e-5
        off.close()
    frame = NutlsOffline(max_frames=32)
    streaming_form = frame.process(clip["mags_in"][:n])
    assert rms(streaming_form[0], want[0]) < 2e-5      # frame 0: both see TA/32
    assert rms(streaming_form[40:], want[40:]) > 1e-4  # later frames: a different function
    frame.close()


@pytest.mark.parametrize("ctfa_mode", ["frame", "causal32"])
def test_block_pipeline_is_independent_of_the_chunk_count(clip, ctfa_mode):
    """The block pipeline (chunks of consecutive frames on their own HIP streams, one bottleneck apart) computes what the
    one-stream block does: 1, 2, 5 (ragged) and 16 chunks, two consecutive blocks so that the carried state and the
    31-frame time-attention history cross a block boundary too.  (Equal up to the summation order inside the conv
    kernels, whose tile shape follows the number of positions per launch: 1e-6 RMS.)"""
    x = clip["mags_in"][:240]
    want = None
    for chunks in (1, 2, 5, 16):
        off = NutlsOffline(max_frames=120, ctfa_mode=ctfa_mode, pipeline=chunks)
        got = off.process(x)
        off.close()
        if want is None:
            want = got
            if ctfa_mode == "frame":
                assert rms(got, clip["mags_out"][:240]) < 2e-5
        else:
            assert rms(got, want) < 1e-6 and float(np.abs(got - want).max()) < 5e-5, chunks


def test_block_pipeline_default_for_long_blocks(clip):
    """1024-frame blocks (the section 8(f).2 workload) run with the automatic chunk count; the result equals the
    frame-by-frame golden outputs of the clip, repeated input and all."""
    x = np.concatenate([clip["mags_in"]] * 5)[:1024]
    off = NutlsOffline(max_frames=1024)
    got = off.process(x)
    one = NutlsOffline(max_frames=1024, pipeline=1)
    ref = one.process(x)
    off.close(); one.close()
    assert rms(got, ref) < 1e-6
    assert rms(got[:249], clip["mags_out"]) < 2e-5


@pytest.mark.parametrize("ctfa_mode", ["frame", "causal32"])
def test_batched_block_mode_equals_one_utterance_handles_and_the_oracle(clip, ctfa_mode):
    """`nutls_create_offline_batch`: the offline forward with a batch dimension (the reference's takes [B, T, ...],
    /root/reference/dnn_model/models/proposed.py:284-625).  Three utterances (different parts of the clip) in ONE handle, ragged block
    lengths with the state of every utterance carried from block to block: each utterance's output equals that of a one-utterance
    handle (same kernels and tilings: <= 1e-6) and the oracle's, in both CTFA modes (causal32 = the training model's 32-frame
    attention, whose history is per utterance); carried states are per utterance; resetting one utterance leaves the others alone."""
    from oracle.nutls_ref import NutlsRef
    frames = clip["mags_in"]
    U, T_max = 3, 24
    starts = [0, 60, 131]
    sizes = [24, 7, 24, 1, 17]          # (sum 73 > 64: the causal32 history wraps; blocks shorter than the 31-frame window, a one-frame block)
    n_total = sum(sizes)
    x = np.stack([frames[s:s + n_total] for s in starts])          # [U, N, 256]
    off = NutlsOffline(max_frames=T_max, utterances=U, ctfa_mode=ctfa_mode)
    outs, t = [], 0
    for n in sizes:
        outs.append(off.process(x[:, t:t + n]))
        assert outs[-1].shape == (U, n, 256)
        t += n
    got = np.concatenate(outs, axis=1)
    ref = NutlsRef(batch=U, ctfa_mode=ctfa_mode)
    want = np.stack([ref.step(x[:, i]).numpy() for i in range(n_total)], axis=1)
    assert rms(got, want) < 2e-5
    for u in range(U):
        one = NutlsOffline(max_frames=T_max, ctfa_mode=ctfa_mode)
        assert rms(got[u], one.process(x[u])) < 1e-6, u
        one.close()
    # carried state: per utterance, equal to the oracle's after the same frames
    h = off.state_get("msfe4_en_h")
    assert h.shape == (U, 21) and rms(h, ref.state["msfe4_en_h"].numpy().reshape(U, 21)) < 2e-5
    p1 = off.state_get("msfe6_ee_prev1")
    assert p1.shape == (U, 256, 64) and rms(p1, ref.state["msfe6_ee_prev1"].numpy()) < 2e-5
    # a new utterance moves into slot 1: the other two continue as if nothing happened
    off.reset_utterance(1)
    y = off.process(np.stack([frames[starts[0] + n_total:starts[0] + n_total + 9], frames[:9], frames[starts[2] + n_total:starts[2] + n_total + 9]]))
    fresh = NutlsOffline(max_frames=T_max, ctfa_mode=ctfa_mode)
    assert rms(y[1], fresh.process(frames[:9])) < 1e-6
    fresh.close()
    cont = np.stack([ref.step(np.stack([frames[starts[0] + n_total + i], frames[i], frames[starts[2] + n_total + i]])).numpy() for i in range(9)], axis=1)
    assert rms(y[0], cont[0]) < 2e-5 and rms(y[2], cont[2]) < 2e-5
    with pytest.raises(ValueError):
        off.process(x[:2, :5])          # (the batch dimension is the handle's)
    off.close()
