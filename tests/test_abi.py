"""The C-ABI library loads and exports every symbol include/memvul_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "memvul_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from memvul_amd import build, binding
    build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    return binding.load_library()


def test_header_and_binding_agree(lib):
    from memvul_amd import binding
    assert _declared_symbols() == sorted(binding.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), f"libmemvul_hip.so does not export {name}"


def test_kernel_class_names(lib):
    names = [lib.mv_kernel_class_name(i).decode() for i in range(12)]
    assert names[1] == "gemm_qkv" and names[5] == "gemm_ffn1_gelu" and len(set(names)) == 12


def test_create_rejects_bad_config_and_missing_gpu(lib):
    from memvul_amd.binding import Engine, MvConfig
    cfg = MvConfig(30522, 1024, 12, 12, 3072, 512, 2, 512, 1e-12, 1024, 8, 8, 0)  # hidden != 768
    h = ctypes.c_void_p()
    assert lib.mv_create(0, ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"768" in lib.mv_last_error(None)
    try:
        import subprocess
        has_gpu = subprocess.run(["/opt/rocm/bin/rocm_agent_enumerator"], capture_output=True, text=True).stdout.count("gfx9") > 0
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(RuntimeError):  # the product path fails loudly without a GPU: no CPU fallback
            Engine(0)


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under memvul_amd/ may import it."""
    pkg = os.path.join(ROOT, "memvul_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, fn)


def test_no_cpp_exception_can_cross_the_abi():
    """include/memvul_hip.h promises that no C++ exception leaves the library: every `int mv_*` entry of engine.hip that has a
    body of its own is a function-try-block whose handler translates to a status (engine.hip on_exception); only one-line
    accessors that cannot throw are exempt."""
    src = open(os.path.join(ROOT, "memvul_amd", "csrc", "engine.hip")).read()
    block = src[src.index('extern "C" {'):src.index('}  // extern "C"')]
    heads = re.findall(r"^int (mv_\w+)\(([^{;]*?)\)\s*(try\s*)?\{(.*)$", block, flags=re.M)
    assert len(heads) >= 30
    unguarded = [name for name, _, guard, rest in heads if not guard and not rest.rstrip().endswith("}")]
    assert not unguarded, unguarded
    assert block.count("catch (...) { return on_exception(") == sum(1 for _, _, guard, _ in heads if guard)


def test_stamp_carries_the_device_code_fingerprint(lib):
    """bench.py keys the counter figures of profiles/pmc_current.json on the stamp's `dev` line = sha256 of the gfx950 code
    object inside the .so: it must describe the binary next to it, and the committed counter file must be readable."""
    import json

    import bench
    from memvul_amd import build

    st = build.read_stamp()
    assert set(st) >= {"src", "cc", "dev"}
    assert st["dev"] == build.device_code_fingerprint() and st["src"] == build.source_fingerprint()
    pmc = json.load(open(bench.PMC_FILE))
    modes = sorted(pmc.get("classes_by_mode", {}))
    assert modes, "profiles/pmc_current.json: one counter pass per compute dtype under classes_by_mode"
    classes, note = bench.load_pmc(modes[0])
    if "dev " + st["dev"] in pmc["lib_stamp"]:
        assert classes and "gemm_ffn2" in classes and classes["gemm_ffn2"]["traffic_bytes"] > 0, note
        assert bench.load_pmc("no such mode")[0] == {}
    else:  # kernels changed since the last counter pass: the bench must say so instead of quoting the old figures
        assert classes == {} and "not reported" in note
