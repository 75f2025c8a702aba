"""Builds the HIP library in-tree:  csrc/*.hip, csrc/*.cpp  ->  libnutls_hip.so  (gfx950 only).

    python -m nunet_amd.build        (or: from nunet_amd.build import build; build())

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the working-tree snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnutls_hip.so")
SOURCES = ["kernels.hip", "megakernel.hip", "stft.hip", "offline.hip", "weights.cpp", "engine.cpp"]
HEADERS = ["nutls_internal.hpp", os.path.join("..", "..", "include", "nutls.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libnutls_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libnutls_hip.so")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
