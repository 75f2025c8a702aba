"""Builds the HIP library in-tree:  csrc/*.hip, csrc/*.cpp  ->  libnutls_hip.so  (gfx950 only).

    python -m nunet_amd.build        (or: from nunet_amd.build import build; build())

hipcc cross-compiles without a GPU.  Every source is compiled to its own object (in parallel; the fused
kernel's 154 specialised ops take minutes, everything else seconds) and only re-compiled when it or a header
changed; the objects and the .so are git-ignored, the .so travels to the GPU box with the working-tree snapshot.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libnutls_hip.so")
SOURCES = ["fused_step.hip", "fused_step_g2.hip", "fused_step_g4.hip", "fused_step_prof.hip", "fused_step_stop.hip", "fused_base.hip", "fused_base_prof.hip", "kernels.hip", "stft.hip", "offline.hip", "weights.cpp", "fused_host.cpp", "engine.cpp"]
if os.environ.get("NUTLS_BUILD_G4_PROF") == "1":      # developer knob: the profiling twin of the 4-stream packed kernel (12 more minutes)
    SOURCES.insert(3, "fused_step_g4_prof.hip")
# (headers are found by scanning the #include "..." lines of every source: _deps)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LAST_BUILD = None      # what the last build() call did: {"mode", "rebuilt": [...], "sources", "seconds", "linked"}


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libnutls_hip.so")
    return hipcc


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path: str, seen=None) -> set:
    """The file and everything it includes with quotes, transitively (paths relative to the including file)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, "r", errors="replace") as f:
        for inc in _INC.findall(f.read()):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _stale_sources(force: bool):
    out = []
    for s in SOURCES:
        o = _obj(s)
        dep = max(os.path.getmtime(d) for d in _deps(os.path.join(CSRC, s)))
        if force or not os.path.exists(o) or os.path.getmtime(o) < dep:
            out.append(s)
    return out


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed: " + " ".join(cmd[-3:]))
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    todo = _stale_sources(force)
    only = os.environ.get("NUTLS_BUILD_ONLY")      # developer knob while iterating on one kernel: re-compile just these (stale objects of the others are linked as they are)
    if only:
        todo = [s for s in todo if s in only.split(",")]
    global LAST_BUILD
    if not todo and not only and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(_obj(s)) for s in SOURCES):
        LAST_BUILD = {"mode": "force" if force else "incremental", "rebuilt": [], "sources": len(SOURCES), "seconds": 0.0, "linked": False}
        print("nutls build: rebuilt 0 of %d sources (every object is newer than its source and headers), library up to date" % len(SOURCES), flush=True)
        return LIB
    import time
    t0 = time.time()
    # (the longest compiles first: the 4-stream packed kernel alone is 12 minutes, the pool must not start it last)
    order = sorted(todo, key=lambda s: {"fused_step_g4.hip": 0, "fused_step_g4_prof.hip": 0, "fused_step_g2.hip": 1}.get(s, 2))
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(lambda s: _run([hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", _obj(s)], verbose), order))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES], verbose)
    LAST_BUILD = {"mode": "force" if force else "incremental", "rebuilt": list(todo), "sources": len(SOURCES), "seconds": round(time.time() - t0, 1), "linked": True}
    print("nutls build: rebuilt %d of %d sources (%s) and linked in %.0f s" % (len(todo), len(SOURCES), ", ".join(todo) or "-", time.time() - t0), flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
