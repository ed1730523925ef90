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
SOURCES = ["coords.hip", "conv.hip", "conv_f16x3.hip", "conv_dma.hip", "conv_lin.hip", "conv_wide.hip", "conv_grid.hip", "conv_win.hip", "conv_wop.hip", "rows.hip", "attn.hip", "input.hip", "panop.hip"]
HEADERS = ["ph_common.h", "conv_h2_common.h", os.path.join("..", "..", "include", "pasco_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


DEV_LIB_PATH = os.path.join(CSRC, "libpascohip_dev.so")


def build_hip(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    """Compile every .hip translation unit and link libpascohip.so. Returns the library path.

    dev=True builds the DEVELOPMENT library instead (-DPH_DEV -> libpascohip_dev.so, objects *.dev.o): the same kernels plus the
    experiment surface the product library does not have - PASCO_* environment switches inside the dispatch, ablation masks,
    shader-clock traces and their extern "C" setters.  Only tools/ loads it (tools/devlib.py)."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    lib_path = DEV_LIB_PATH if dev else LIB_PATH
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".dev.o" if dev else ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *FLAGS, *(["-DPH_DEV"] if dev else []), "-c", src, "-o", obj])

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
    if force or jobs or _stale(lib_path, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
    return lib_path


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, dev="--dev" in sys.argv))
