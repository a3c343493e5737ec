"""Build libmemvul_hip.so (gfx950) in-tree with hipcc.  No torch involved."""
from __future__ import annotations

import os
import shutil
import subprocess
from struct import error as struct_error
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmemvul_hip.so")
# the same sources with -DMEMVUL_DEV_SWITCHES: the development A/B knobs (MEMVUL_GEMM_TILE, MEMVUL_SHORT_VLO, MEMVUL_RASTER, MEMVUL_GN_MAX,
# MEMVUL_NUM_CU; engine.hip mv_create) exist only in this build — GPU tests that force a kernel path at test sizes and the A/B scripts load it
LIB_PATH_DEV = os.path.join(LIB_DIR, "libmemvul_hip_dev.so")
DEV_FLAGS = ("-DMEMVUL_DEV_SWITCHES",)
SOURCES = ["engine.hip"]
HEADERS = ["common.h", "gemm.h", "gemm_pp.h", "attention.h", "attention_v2.h", "misc_kernels.h", "match_topk.h", os.path.join(ROOT, "include", "memvul_hip.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm at /opt/rocm)")


STAMP_PATH = LIB_PATH + ".stamp"


def _paths(dev: bool):
    lib = LIB_PATH_DEV if dev else LIB_PATH
    return lib, lib + ".stamp"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _compiler_id():
    try:
        return subprocess.run([hipcc_path(), "--version"], capture_output=True, check=True).stdout
    except Exception:  # no compiler here (a GPU box without the ROCm dev tools)
        return None


def source_fingerprint(extra_flags=()) -> str:
    """sha256 of the flags and the CONTENT of every source / header (not mtimes — a checkout or a copy to another box
    rewrites those)."""
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join([ARCH, *FLAGS, *extra_flags]).encode())
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h_ if os.path.isabs(h_) else os.path.join(CSRC, h_) for h_ in HEADERS]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_fingerprint(extra_flags=()) -> str:
    """What the binary depends on, as the two lines of the stamp file: `src <sha256 of flags + sources>` and
    `cc <sha256 of the compiler's --version>`.  bench.py keys its counter figures on the whole stamp."""
    import hashlib

    cc = _compiler_id()
    return "src %s\ncc %s" % (source_fingerprint(extra_flags), hashlib.sha256(cc).hexdigest() if cc is not None else "unknown")


def device_code_fingerprint(lib_path: str = None) -> str:
    """sha256 of the gfx950 code object(s) embedded in the shared library (the clang offload bundle inside .hip_fatbin): what the
    GPU executes.  Host-only edits of engine.hip leave it unchanged, any kernel change moves it — bench.py keys the counter
    figures of profiles/pmc_current.json on this line of the stamp."""
    import hashlib
    import struct

    data = open(lib_path or LIB_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    h = hashlib.sha256()
    found = 0
    i = data.find(magic)
    while i >= 0:
        if i + 32 > len(data):
            raise RuntimeError("truncated offload bundle header")
        (n,) = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        if n > 64:  # a handful of targets at most: anything else is not a bundle header we understand (e.g. a compressed one)
            raise RuntimeError(f"offload bundle with {n} entries: not a plain clang offload bundle")
        for _ in range(n):
            if off + 24 > len(data):
                raise RuntimeError("truncated offload bundle entry table")
            o, size, tsz = struct.unpack_from("<QQQ", data, off)
            off += 24
            if tsz > 256 or off + tsz > len(data) or i + o + size > len(data):  # never trust offsets read from the file
                raise RuntimeError("offload bundle entry points outside the file")
            triple = data[off:off + tsz]
            off += tsz
            if ARCH.encode() in triple:
                h.update(data[i + o:i + o + size])
                found += 1
        i = data.find(magic, i + len(magic))
    if not found:
        raise RuntimeError(f"no {ARCH} code object found in {lib_path or LIB_PATH}")
    return h.hexdigest()


def read_stamp(dev: bool = False) -> dict:
    """The stamp next to the library as a dict: src / cc (what it was built from) and dev (the device code it contains)."""
    with open(_paths(dev)[1]) as f:
        return dict(line.split(" ", 1) for line in f.read().strip().splitlines() if " " in line)


def is_stale(extra_flags=(), dev: bool = False) -> bool:
    """True when the library must be rebuilt: no binary / stamp, other sources or flags, or — where a compiler exists to
    compare with — another compiler.  On a box WITHOUT hipcc only the source line is compared (the binary that travelled with
    the tree is then the one to use; ADVICE r2: the old single-hash stamp could never match there)."""
    lib_path, stamp_path = _paths(dev)
    if dev:
        extra_flags = tuple(extra_flags) + DEV_FLAGS
    if not os.path.exists(lib_path) or not os.path.exists(stamp_path):
        return True
    with open(stamp_path) as f:
        have = dict(line.split(" ", 1) for line in f.read().strip().splitlines() if " " in line)
    want = dict(line.split(" ", 1) for line in build_fingerprint(extra_flags).splitlines())
    if have.get("src") != want["src"]:
        return True
    return want["cc"] != "unknown" and have.get("cc") != want["cc"]


def build(force: bool = False, verbose: bool = True, extra_flags=(), dev: bool = False) -> str:
    """Compile the HIP library for gfx950; returns the .so path.  dev: the -DMEMVUL_DEV_SWITCHES build (libmemvul_hip_dev.so)."""
    if not force and not is_stale(extra_flags, dev):
        return _paths(dev)[0]
    lib_path, stamp_path = _paths(dev)
    if dev:
        extra_flags = tuple(extra_flags) + DEV_FLAGS
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", *FLAGS,  # -Wno-unused-value: hipError_t of calls checked by launch_check
        *extra_flags,
        *[os.path.join(CSRC, s) for s in SOURCES],
        "-o", lib_path + ".tmp",
    ]
    if verbose:
        print("[memvul_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # the stamp describes the NEW binary and is written from it before either file is moved into place; a fatbin this parser does
    # not understand (compressed bundle, no gfx950 entry) yields `dev unknown` — bench.load_pmc then compares whole stamps — instead
    # of a successful compile left without a stamp (and rebuilt, and failing again, on every load)
    try:
        code = device_code_fingerprint(lib_path + ".tmp")
    except (RuntimeError, OSError, ValueError, struct_error) as e:
        code = "unknown"
        if verbose:
            print(f"[memvul_amd.build] device-code fingerprint unavailable ({e}); stamp carries `dev unknown`", flush=True)
    with open(stamp_path + ".tmp", "w") as f:
        f.write(build_fingerprint(extra_flags) + "\ndev " + code + "\n")
    os.replace(lib_path + ".tmp", lib_path)
    os.replace(stamp_path + ".tmp", stamp_path)
    return lib_path


def build_all(force: bool = False, verbose: bool = True):
    """Both builds, compiled side by side (two hipcc processes)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:
        futs = [ex.submit(build, force, verbose, (), d) for d in (False, True)]
        return [f.result() for f in futs]


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv)))
