"""Build the CPU oracle (test infrastructure) into oracle/libpasco_oracle.so with gcc + OpenMP.

No -march=native: the .so built in the dev container travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pasco_oracle.c")
LIB = os.path.join(HERE, "libpasco_oracle.so")
HDR = os.path.join(HERE, "..", "include", "pasco_hip.h")


def build_oracle(force: bool = False) -> str:
    stale = (not os.path.exists(LIB)) or any(
        os.path.getmtime(p) > os.path.getmtime(LIB) for p in (SRC, HDR))
    if force or stale:
        cmd = ["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-std=gnu11", "-Wall", "-shared", "-fPIC",
               SRC, "-o", LIB, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("oracle build failed")
    return LIB


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
