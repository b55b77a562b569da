"""Build recipe for libmgacbam.so (hand-written HIP for gfx950, plain C ABI -- include/mgacbam.h).

The library is built IN-TREE with hipcc (no torch C++ extension: the ABI carries no torch types, and the
image's hipcc (ROCm 7.2) differs from torch's bundled HIP (7.0); at run time the .so resolves
``libamdhip64.so.7`` to the copy torch has already loaded, so both share one HIP runtime and streams).
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmgacbam.so")
SOURCES = ["mgacbam_api.hip"]
ARCH = "gfx950"


def dependencies():
    """Everything the library is compiled from: every file under csrc/ (all .cuh are included by mgacbam_api.hip) + the public header."""
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cuh", ".h"))]
    return deps + [os.path.join(ROOT, "include", "mgacbam.h")]


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: libmgacbam.so cannot be built on this machine")
    return p


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in dependencies())


def build(force: bool = False, verbose: bool = False, resource_log: str | None = None, defines=(), out: str | None = None) -> str:
    """Compile csrc/*.hip for gfx950 into mga_yolo_amd/libmgacbam.so; returns the library path."""
    if not force and not is_stale() and not out:
        return LIB
    cmd = [hipcc_path(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-shared", "-fPIC",
           "-Wno-pass-failed", "-o", out or LIB] + list(defines) + [os.path.join(CSRC, s) for s in SOURCES]
    if resource_log:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if resource_log:
        with open(resource_log, "w") as f:
            f.write(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    return out or LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
