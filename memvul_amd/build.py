"""Build libmemvul_hip.so (gfx950) in-tree with hipcc.  No torch involved."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmemvul_hip.so")
SOURCES = ["engine.hip"]
HEADERS = ["common.h", "gemm.h", "gemm_pp.h", "attention.h", "attention_v2.h", "misc_kernels.h", os.path.join(ROOT, "include", "memvul_hip.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm at /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    """Compile the HIP library for gfx950; returns the .so path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value",  # hipError_t of calls checked by launch_check
        *extra_flags,
        *[os.path.join(CSRC, s) for s in SOURCES],
        "-o", LIB_PATH + ".tmp",
    ]
    if verbose:
        print("[memvul_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
