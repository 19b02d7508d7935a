"""include/ptmi.h consumed from plain C99 -- the language cgo compiles it with: tests/c_consumer/
shim_check.c is built with gcc against libptmi.so and makes the calls the Go shim makes."""
import os
import subprocess

import pytest

from pathtrace_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "shim_check")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_consumer", "shim_check.c"), "-o", exe, "-L", libdir, "-lptmi", "-lm",
                           "-Wl,-rpath," + libdir])
    return exe


def test_header_is_plain_c_and_host_calls_work(tmp_path):
    out = subprocess.run([_build(tmp_path), "host"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "shim_check host ok" in out.stdout


@pytest.mark.gpu
def test_shim_call_sequence_on_the_gpu(tmp_path):
    """build -> closest hit -> any hit -> scene -> render from C, hand-derived answers (no oracle involved)."""
    out = subprocess.run([_build(tmp_path), "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "shim_check gpu ok" in out.stdout
