"""Build the in-tree HIP library ``hept_amd/csrc/libhept_hip.so`` for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libhept_hip.so")


def build(force: bool = False, jobs: int = 4) -> str:
    """Run ``make`` in ``csrc/`` (incremental); returns the path of the shared library."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc build of libhept_hip.so failed:\n" + proc.stdout + proc.stderr)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("make finished but " + LIB_PATH + " is missing")
    return LIB_PATH


if __name__ == "__main__":
    print(build())
