"""CPU-side checks of the drop-in boundary: the built library loads, exports every symbol
``include/nutls.h`` declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import nunet_amd
from nunet_amd import runner
from nunet_amd.build import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return runner.load_library()


def test_header_and_library_agree(lib):
    hdr = open(os.path.join(ROOT, "include", "nutls.h")).read()
    declared = set(re.findall(r"\b(nutls_[a-z_]+)\s*\(", hdr))
    assert declared == set(runner.ABI_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.nutls_version()


def test_bad_arguments_are_reported(lib):
    h = ctypes.c_void_p()
    assert lib.nutls_create(None, 0, 0, 1, 0, ctypes.byref(h)) == -1
    assert b"null" in lib.nutls_last_error()
    assert lib.nutls_create(b"x", 1, 7, 1, 0, ctypes.byref(h)) == -1      # unknown variant
    assert lib.nutls_state_count(None) == -1
    # the widened entry points (STFT front end, offline mode) validate before touching the device
    assert lib.nutls_create_offline(b"x", 1, 0, 0, ctypes.byref(h)) == -1       # max_frames < 1
    assert b"max_frames" in lib.nutls_last_error()
    assert lib.nutls_process_block_host(None, None, None, 1) == -1
    assert lib.nutls_enhance_hop_host(None, None, None, 0) == -1
    assert lib.nutls_stft_hop(None, None, None) == -1
    # page-locked host buffers: zero bytes is an error, freeing NULL or a pointer the library did not hand out is a no-op
    assert lib.nutls_host_alloc(0) is None
    assert b"nutls_host_alloc" in lib.nutls_last_error()
    lib.nutls_host_free(None)
    buf = (ctypes.c_float * 4)()
    lib.nutls_host_free(ctypes.addressof(buf))
    assert lib.nutls_profile_production(None, None, 0, 1, 1) == -1


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        nunet_amd.NutlsEngine(batch=1)
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        nunet_amd.NutlsOffline(max_frames=8)
    with pytest.raises(RuntimeError, match="hipHostMalloc"):      # (and page-locked memory needs the HIP runtime's device too)
        nunet_amd.host_alloc((2, 256))


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_fused_host_entry_points_validate_their_arguments(lib):
    """The host-only entry points of the fused kernel's plan (no GPU needed) know both variants and nothing else."""
    assert lib.nutls_fused_num_ops(0) == 154 and lib.nutls_fused_num_ops(1) == 154 and lib.nutls_fused_num_ops(2) == 0
    assert lib.nutls_fused_blob_floats(0) > 0 and lib.nutls_fused_blob_floats(1) > 0 and lib.nutls_fused_blob_floats(-1) == 0
    name, fl = ctypes.c_char_p(), ctypes.c_double()
    assert lib.nutls_fused_op_info(1, 8, ctypes.byref(name), ctypes.byref(fl)) == 0
    assert name.value == b"msfe6_en_ddb" and fl.value > 0          # the baseline plan's first dilated-dense op
    assert lib.nutls_fused_op_info(0, 8, ctypes.byref(name), ctypes.byref(fl)) == 0 and name.value == b"msfe6_en_lstm"
    assert lib.nutls_fused_op_info(0, 154, ctypes.byref(name), ctypes.byref(fl)) == -1
    assert lib.nutls_fused_op_info(3, 0, ctypes.byref(name), ctypes.byref(fl)) == -1
    assert lib.nutls_offline_set_pipeline(None, 2) == -1
    assert lib.nutls_offline_set_ctfa_mode(None, 0) == -1


def test_int8_container_round_trip():
    """`write_blob(..., int8_convs=True)` stores the conv kernels the way the reference's export does (symmetric int8 per
    output channel, tensors below 1024 elements and the dilated-dense blocks' kernels stay float); `parse_blob` gives
    back exactly q * scale, and the library's own parser accepts the container."""
    import numpy as np
    from nunet_amd.weights import parse_blob, quantize_conv_kernels, synthetic_weights, write_blob
    w = synthetic_weights("baseline", seed=3, bias_std=0.1, affine_jitter=0.1)
    q = quantize_conv_kernels(w)
    blob = write_blob(w, int8_convs=True)
    back, raw = parse_blob(blob), parse_blob(blob, dequantize=False)
    assert set(back) == set(w)
    n_q = 0
    for name, arr in w.items():
        if isinstance(q[name], tuple):
            n_q += 1
            qi, sc = q[name]
            assert qi.dtype == np.int8 and sc.shape == (arr.shape[0],) and np.abs(qi).max() <= 127
            assert np.array_equal(raw[name][0], qi) and np.array_equal(raw[name][1], sc)
            assert np.array_equal(back[name], qi.astype(np.float32) * sc.reshape(-1, 1, 1, 1))
            assert np.abs(back[name] - arr).max() <= 0.5 * sc.max() * 1.0001            # half a quantisation step
        else:
            assert np.array_equal(back[name], np.asarray(arr, np.float32)), name
            assert "ddb" in name or not name.endswith(".w") or np.asarray(arr).ndim != 4 or np.asarray(arr).size < 1024
    assert n_q == 128                                            # every encoder / decoder conv kernel of the network


def test_baseline_tflite_export_rekeys_to_the_container_names():
    """The 'nutls' export (converter_nunet_tls.py:1538-1552) is not shipped, so the importer is exercised on a constants
    table built from synthetic baseline weights under the tensor names TF-Lite gives them (Keras layer / conv2d_N / ...,
    SURVEY A.9): a dilated-dense block is one Sequential with two convs whose biases only differ by the conv2d_N component."""
    import numpy as np
    from nunet_amd.weights import synthetic_weights
    from tools.convert_tflite_weights import rekey
    from tools.tflite_reader import TensorInfo

    want = synthetic_weights("baseline", seed=11, bias_std=0.1, affine_jitter=0.1)
    consts, n = {}, [0, 0, 0]

    def add(name, arr):
        a = np.asarray(arr, np.float32)
        consts[name] = TensorInfo(len(consts), name, a.shape, np.float32, 0, np.zeros(0, np.float32), np.zeros(0, np.int64), 0, a)

    layers = sorted({k.rsplit(".", 1)[0] for k in want})
    for L in layers:
        roles = {k.rsplit(".", 1)[1] for k in want if k.rsplit(".", 1)[0] == L}
        keras = "conv2d" if L == "out_conv" else L
        if roles >= {"wg", "w1"}:                       # dilated-dense block: grouped conv, 1x1 conv, LN, PReLU
            g = want[L + ".w1"].shape[0]
            for w, b, shape in ((".wg", ".bg", None), (".w1", ".b1", (g, 1, 1, g))):
                n[0] += 1
                add("%s/conv2d_%d/Conv2D" % (keras, n[0]), want[L + w] if shape is None else want[L + w].reshape(shape))
                add("%s/conv2d_%d/BiasAdd/ReadVariableOp" % (keras, n[0]), want[L + b])
        elif roles >= {"w1", "w2"}:                     # CTFA gate perceptrons
            for w, b in ((".w1", ".b1"), (".w2", ".b2")):
                n[0] += 1
                add("%s/conv1d_%d/Conv1D" % (keras, n[0]), want[L + w])
                add("%s/conv1d_%d/BiasAdd/ReadVariableOp" % (keras, n[0]), want[L + b])
        else:
            n[0] += 1
            tr = "upsampling" in L
            add("%s/conv2d_%s%d/%s" % (keras, "transpose_" if tr else "", n[0], "conv2d_transpose" if tr else "Conv2D"), want[L + ".w"])
            add("%s/conv2d_%s%d/BiasAdd/ReadVariableOp" % (keras, "transpose_" if tr else "", n[0]), want[L + ".b"])
        if "gamma" in roles:
            n[1] += 1
            add("%s/layer_normalization_%d/batchnorm/mul/ReadVariableOp" % (keras, n[1]), want[L + ".gamma"])
            add("%s/layer_normalization_%d/batchnorm/ReadVariableOp" % (keras, n[1]), want[L + ".beta"])
            add("%s/layer_normalization_%d/batchnorm/add/y" % (keras, n[1]), np.float32(1e-8))
        if "alpha" in roles:
            n[2] += 1
            add("%s/p_re_lu_%d/Neg/ReadVariableOp" % (keras, n[2]), want[L + ".alpha"])

    class Model:
        def constants(self):
            return consts

    got = rekey(Model())
    assert set(got) == set(want)
    for k, v in want.items():
        assert np.array_equal(np.asarray(got[k].data).reshape(np.asarray(v).shape), v), k
    assert got["msfe6_en_ddb_3.w1"].shape == (16, 16) and got["ddb_6.wg"].shape == (32, 2, 3, 6)


def test_plan_cost_constants_match_the_measurement():
    """engine.cpp picks the fused plan by rounds of workgroups x step time of the plan; the step-time RATIOS are constants in the source
    (fused_setup: t_plan).  They are tied to profiles/plan_cost_model.json (tools/gpu_plan_cost.py, one box, back to back): more than 3 %
    apart fails -- re-measure after a kernel change, then update the constants."""
    import json
    import re
    src = open(os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd", "csrc", "engine.cpp")).read()
    m = re.search(r"t_plan\[5\]\s*=\s*\{([^}]*)\}", src)
    assert m, "t_plan constants not found in engine.cpp"
    consts = [float(x) for x in m.group(1).split(",")]
    rec = json.load(open(os.path.join(ROOT, "profiles", "plan_cost_model.json")))
    for g in (1, 2, 4):
        meas = rec["ratio_to_one_stream"][str(g)]
        assert abs(consts[g] / meas - 1.0) <= 0.03, (g, consts[g], meas)
