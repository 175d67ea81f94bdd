"""The C-ABI library builds, loads on a machine without a GPU, and exports exactly the entry points that
include/sdlt_kernels.h declares (no compute calls here)."""
import ctypes
import os
import re

from sd_lora_trainer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sdlt_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdlt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/sdlt_kernels.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding table and header disagree"
    assert lib.sdlt_abi_version() == 1


def test_struct_layouts_match():
    lib = _lib.load()
    sizes = _lib.struct_sizes()
    # every `typedef struct` of the header has an entry (16 at the time of writing), and nothing beyond the table answers
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdlt_kernels.h")).read()
    assert len(sizes) == len(set(re.findall(r"typedef struct (sdlt_\w+)", header))), "a struct of the header is missing from sdlt_struct_size / _lib.struct_sizes"
    for which, (name, size) in enumerate(sizes):
        assert lib.sdlt_struct_size(which) == size, name
    assert lib.sdlt_struct_size(len(sizes)) == -1


def test_errors_are_reported_not_swallowed():
    lib = _lib.load()
    p = _lib.GemmParams()          # all zero: invalid shape -> negative code + message, no launch attempted
    rc = lib.sdlt_gemm_bf16(ctypes.byref(p), None)
    assert rc < 0 and b"sdlt_gemm_bf16" in lib.sdlt_last_error()


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.KernelLibraryError as e:
        assert "no CPU / PyTorch fallback" in str(e)
    else:
        raise AssertionError("loading a missing kernel library must raise")
