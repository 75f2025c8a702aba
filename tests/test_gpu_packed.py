"""GPU parity of the PACKED plans of the fused kernel (two / four streams per workgroup, csrc/fused_step_g2.hip / _g4.hip, BASELINE configs[3] / [4]):
the same checks the one-stream plan passes -- goldens of the shipped graph (outputs and every state tensor), oracle B on synthetic
streams -- plus what packing adds: a stream's results do not depend on its slot in the workgroup or on its partner, both plans
compute the same function, and the carried partial sums survive state edits.  Reference semantics:
/root/reference/dnn_model/converter_proposed.py:188-867 (one step of the signature), one stream per call there."""
import os

import numpy as np
import pytest

import nunet_amd.topology as T
from conftest import GOLDEN
from nunet_amd import NutlsEngine
from oracle.nutls_ref import NutlsRef

pytestmark = pytest.mark.gpu
TIGHT_RMS = 2e-5          # fp32 on both sides; differences are summation order only (north-star tolerance: 1e-3)


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.fixture(scope="module")
def clip():
    return np.load(os.path.join(GOLDEN, "clip_4s.npz"))


def test_plan_choice_follows_the_stream_count():
    """More streams than CUs: a packed plan by default, chosen by rounds x step time; an odd count or NUTLS_FUSED_STREAMS=1: one stream per
    workgroup; an explicit request for a plan the batch cannot run is an error."""
    # (the choice minimises rounds x step time of the plan: 300 streams are one round of pairs instead of two rounds of single streams,
    #  768 three rounds of single streams rather than one round of 192 four-stream workgroups at 3.56x; 1024 / 1536 / 2048: 2 / 3 / 4 rounds of
    #  pairs at 1.66x rather than 1 / 2 / 2 rounds of fours -- since round 6's cache policy the pairs win there: tools/exp/plan_ab.py)
    for B, want in ((1, 1), (256, 1), (300, 2), (511, 1), (512, 2), (768, 1), (1022, 2), (1024, 2), (1536, 2), (2048, 2)):
        eng = NutlsEngine(batch=B)
        assert eng.streams_per_workgroup == want, (B, eng.streams_per_workgroup)
        eng.close()
    eng = NutlsEngine(batch=512, streams_per_workgroup=1)
    assert eng.streams_per_workgroup == 1
    eng.close()
    eng = NutlsEngine(batch=6, streams_per_workgroup=2)
    assert eng.streams_per_workgroup == 2
    eng.close()
    with pytest.raises(ValueError):                              # an explicit request that cannot be met fails loudly (it never becomes another plan)
        NutlsEngine(batch=5, streams_per_workgroup=2)
    with pytest.raises(ValueError):
        NutlsEngine(batch=6, streams_per_workgroup=3)
    eng = NutlsEngine(batch=8, streams_per_workgroup=4)
    assert eng.streams_per_workgroup == 4
    eng.close()
    # a container the fused kernel cannot run (float conv kernels): the library's own choice falls to the per-layer kernels, an explicit
    # plan request says why it cannot be met (include/nutls.h, nutls_create_plan)
    from nunet_amd.weights import synthetic_weights, write_blob
    float_blob = write_blob(synthetic_weights("lstm", seed=3))
    eng = NutlsEngine(batch=2, weights=float_blob)
    assert eng.mode != "fused"
    eng.close()
    with pytest.raises(RuntimeError, match="fused kernel"):
        NutlsEngine(batch=2, weights=float_blob, streams_per_workgroup=1)


@pytest.mark.parametrize("G", [2, 4])
def test_packed_golden_clip_and_every_state_tensor(clip, G):
    """G copies of the golden clip, a fraction of a second apart, on ONE workgroup: outputs of 64 frames and all 130 state tensors of
    every slot against the oracle-A goldens (slot s runs 11 s frames behind: its goldens are checked when IT has seen 3 / 64 frames)."""
    eng = NutlsEngine(batch=G, streams_per_workgroup=G)
    assert eng.streams_per_workgroup == G
    lag = [11 * s for s in range(G)]
    zeros = np.zeros(256, np.float32)
    outs = [[] for _ in range(G)]
    for i in range(64 + lag[-1]):
        x = [clip["mags_in"][i - lag[s]] if 0 <= i - lag[s] < 64 else zeros for s in range(G)]
        for s in range(1, G):
            if i == lag[s]:
                eng.reset(s)          # slot s starts its clip from the all-zero state
        out = eng.step(np.stack(x))
        for s in range(G):
            seen = i + 1 - lag[s]
            if 1 <= seen <= 64:
                outs[s].append(out[s])
            if seen in (3, 64):
                st = np.load(os.path.join(GOLDEN, "state_f%d.npz" % seen))
                for base, shp in T.state_specs():
                    k_in = base if len(shp) == 1 else base.format("prev")
                    k_gold = base if len(shp) == 1 else base.format("cur")
                    got = eng.state_get(k_in)[s].reshape(-1)
                    np.testing.assert_allclose(got, st[k_gold].reshape(-1), rtol=1e-4, atol=1e-4, err_msg="slot %d %s" % (s, k_gold))
    for s in range(G):
        assert rms(np.stack(outs[s]), clip["mags_out"][:64]) < TIGHT_RMS, s
    eng.close()


@pytest.mark.parametrize("G", [2, 4])
def test_packed_vs_oracle_and_slot_independence(G):
    """512 synthetic streams (the smallest handle that picks a packed plan by itself), 6 frames: every output against oracle B for
    the first 16 streams; copies of the same input stream give bit-identical results in any slot, next to any partners."""
    B, steps = 512, 6
    rng = np.random.default_rng(7)
    base = (0.25 * np.abs(rng.standard_normal((steps, 16, 256)))).astype(np.float32)
    idx = rng.integers(0, 16, size=B)
    idx[:16] = np.arange(16)
    idx[16:48] = np.repeat(np.arange(16), 2)[::-1]          # the same stream in both slots of a workgroup, and in slot 0 / slot 1 of others
    eng, ref = NutlsEngine(batch=B, streams_per_workgroup=G), NutlsRef(batch=16)
    assert eng.streams_per_workgroup == G
    for s in range(steps):
        out = eng.step(np.ascontiguousarray(base[s][idx]))
        want = ref.step(base[s]).numpy()
        assert np.isfinite(out).all()
        assert rms(out[:16], want) < TIGHT_RMS, s
        for j in range(16):
            same = out[idx == j]
            assert np.array_equal(same, np.broadcast_to(same[0], same.shape)), (s, j)
    for name in ("msfe6_ee_prev1", "msfe6_ee_prev2", "msfe5_de_prev1", "msfe4_dd3_prev2", "msfe3_de_prev1", "state_c", "msfe6_de_h"):
        a, b = eng.state_get(name)[:16].reshape(16, -1), ref.state[name].numpy().reshape(16, -1)
        assert rms(a, b) < 1e-4 * max(1.0, float(np.abs(b).max())), name
    eng.close()


@pytest.mark.parametrize("G", [2, 4])
def test_both_plans_compute_the_same_function(clip, G):
    """The same 8 streams of the real clip through the one-stream plan and a packed plan: equal up to the summation order of the
    layers whose tiling differs (K split, 32x32 instead of 16x16 tiles)."""
    frames = clip["mags_in"]
    a, b = NutlsEngine(batch=8, streams_per_workgroup=1), NutlsEngine(batch=8, streams_per_workgroup=G)
    assert (a.streams_per_workgroup, b.streams_per_workgroup) == (1, G)
    for i in range(40):
        x = np.stack([frames[(i + 17 * s) % 249] for s in range(8)])
        ya, yb = a.step(x), b.step(x)
        assert rms(ya, yb) < 2e-6, i
    for name in ("msfe6_ed_prev6", "msfe5_ee_prev2", "msfe6_dd_prev1", "msfe3_en_h"):
        assert rms(a.state_get(name), b.state_get(name)) < 1e-5, name
    a.close()
    b.close()


@pytest.mark.parametrize("G", [2, 4])
def test_packed_carried_sums_follow_state_edits(clip, G):
    """nutls_state_set on a conv-input state of the packed plan: the library rebuilds the carried partial sums (per-stream layout of the
    packed tilings) before the next step -- the continuation equals that of a handle that never was interrupted."""
    frames = clip["mags_in"]
    a, b = NutlsEngine(batch=4, streams_per_workgroup=G), NutlsEngine(batch=4, streams_per_workgroup=G)
    x = lambda i: np.stack([frames[(i + 31 * s) % 249] for s in range(4)])
    for i in range(5):
        a.step(x(i))
        b.step(x(i))
    for name in ("msfe6_ee_prev1", "msfe6_ee_prev2", "msfe5_ee_prev1", "msfe5_de_prev1", "msfe6_de_prev2", "msfe4_ee3_prev4"):
        b.state_set(name, b.state_get(name))          # same values: marks the sums stale
    for i in range(5, 9):
        assert rms(a.step(x(i)), b.step(x(i))) < 1e-6, i          # (the rebuilt sums are added up in another order than the kernel's)
    a.close()
    b.close()


@pytest.mark.parametrize("G", [2, 4])
def test_state_moves_between_plans(clip, G):
    """The ABI's state tensors mean the same on every plan: an utterance started on a one-stream handle continues on a packed handle after
    all 130 tensors were copied over (nutls_state_get / nutls_state_set; the packed handle rebuilds its carried partial sums, whose
    layout is the plan's own), and back."""
    frames = clip["mags_in"]
    B = 4
    x = lambda i: np.stack([frames[(i + 23 * s) % 249] for s in range(B)])
    ref = NutlsEngine(batch=B, streams_per_workgroup=1)
    a = NutlsEngine(batch=B, streams_per_workgroup=1)
    b = NutlsEngine(batch=B, streams_per_workgroup=G)
    for i in range(6):
        ref.step(x(i))
        a.step(x(i))
    for base, shp in T.state_specs():
        name = base if len(shp) == 1 else base.format("prev")
        b.state_set(name, a.state_get(name))
    for i in range(6, 12):
        assert rms(ref.step(x(i)), b.step(x(i))) < 2e-6, i
    for base, shp in T.state_specs():
        name = base if len(shp) == 1 else base.format("prev")
        a.state_set(name, b.state_get(name))
    for i in range(12, 16):
        assert rms(ref.step(x(i)), a.step(x(i))) < 2e-6, i
    for e in (ref, a, b):
        e.close()


def test_packed_extreme_inputs_and_full_clip(clip):
    """Silence, a single huge bin, denormal-scale noise and the real clip side by side in ONE workgroup of the 4-stream plan: LayerNorm's
    eps path and PReLU's negative side next to ordinary data; the clip's 249 frames in slot 3 against the goldens."""
    B = 4
    eng, ref = NutlsEngine(batch=B, streams_per_workgroup=4), NutlsRef(batch=B)
    assert eng.streams_per_workgroup == 4
    outs = []
    for i in range(249):
        x = np.zeros((B, 256), np.float32)
        x[1, 17] = 1.0e4
        x[2] = 1e-20
        x[3] = clip["mags_in"][i]
        out = eng.step(x)
        assert np.isfinite(out).all()
        outs.append(out[3])
        if i < 3:
            want = ref.step(x).numpy()
            scale = max(1.0, float(np.abs(want).max()))
            assert rms(out, want) < 1e-4 * scale
    assert rms(np.stack(outs), clip["mags_out"]) < TIGHT_RMS
    eng.close()
