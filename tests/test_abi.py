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


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        nunet_amd.NutlsEngine(batch=1)
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        nunet_amd.NutlsOffline(max_frames=8)


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
