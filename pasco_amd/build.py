"""In-tree build of libpascohip.so (hipcc, gfx950 only).

The shared object is written next to the sources (pasco_amd/csrc/libpascohip.so) so that it
travels with the repository snapshot to the GPU box; nothing is installed into site-packages.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libpascohip.so")
SOURCES = ["coords.hip", "conv.hip", "conv_f16x3.hip", "conv_dma.hip", "conv_lin.hip", "conv_wide.hip", "conv_win.hip", "rows.hip", "attn.hip", "input.hip", "panop.hip"]
HEADERS = ["ph_common.h", "conv_h2_common.h", os.path.join("..", "..", "include", "pasco_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip translation unit and link libpascohip.so. Returns the library path."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs])
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
