"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from pathtrace_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "ptmi_closest_hit" in syms and "ptmi_build_bvh" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libptmi.so does not export: {missing}"


def test_python_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_version_and_error_paths_without_compute():
    lib = _lib.load()
    assert b"ptmi" in lib.ptmi_version()
    # null out-pointer is rejected with a message, nothing aborts
    assert lib.ptmi_create(0, None) == -1
    assert b"null" in lib.ptmi_last_error(None)
    # host-only entry points reject bad input with error codes
    h = C.c_void_p()
    assert lib.ptmi_host_bvh_build(None, 3, None, 1, None, 4, 42, C.byref(h)) == -1
    assert lib.ptmi_host_bvh_sizes(None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libptmi.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pathtrace_amd import Context, PtmiError
    with pytest.raises(PtmiError) as e:
        Context(0)
    assert e.value.code == -2


def test_product_and_tools_never_load_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may load it."""
    import glob
    offenders = []
    for path in glob.glob(os.path.join(ROOT, "pathtrace_amd", "**", "*"), recursive=True) + glob.glob(os.path.join(ROOT, "tools", "*")):
        if not os.path.isfile(path) or not path.endswith((".py", ".sh", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
            continue
        text = open(path, errors="replace").read()
        if re.search(r"oracle_lib|liborc|orc_bvh|orc_trace|/oracle\b|\boracle/", text):
            offenders.append(os.path.relpath(path, ROOT))
    assert not offenders, offenders
