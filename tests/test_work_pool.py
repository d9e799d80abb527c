"""The helper pool behind the dB finish and the group's queue copies (csrc/work_pool.h), without a
GPU: a C++ driver runs thousands of jobs through it and counts every item; once plainly and, where
the toolchain links it, once under ThreadSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "pool_test.cpp")
INC = os.path.join(ROOT, "libcoolmic-dsp_amd", "csrc")


def _build(tmp_path, name, extra):
    exe = tmp_path / name
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-pthread", "-Wall", "-Wextra", "-I", INC, SRC,
                        "-o", str(exe)] + extra, capture_output=True, text=True)
    return exe, r


def test_pool_counts_every_item_once(tmp_path):
    exe, r = _build(tmp_path, "pool_test", [])
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe), "3000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "pool ok" in out.stdout, out.stdout + out.stderr


def test_pool_under_thread_sanitizer(tmp_path):
    exe, r = _build(tmp_path, "pool_tsan", ["-fsanitize=thread"])
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer in this toolchain: " + r.stderr[-200:])
    out = subprocess.run([str(exe), "600"], capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in dict(os.environ, TSAN_OPTIONS="halt_on_error=1").items()
                              if k != "LD_PRELOAD"})      # (another sanitizer's runtime, when the suite runs under one)
    assert out.returncode == 0 and "pool ok" in out.stdout, out.stdout + out.stderr[-2000:]
