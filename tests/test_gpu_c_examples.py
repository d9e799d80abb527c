"""The boundary from plain C: the example programs are compiled with gcc against include/
and linked to libcoolmic-dsp-hip.so, then run on the GPU; their output is checked against
the SURVEY 8(c) vectors and the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libcoolmic-dsp_amd", "lib")


def _build_and_run(tmp_path, name, *args):
    exe = tmp_path / name
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-O2",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                    "-L", LIBDIR, "-lcoolmic-dsp-hip", "-Wl,-rpath," + LIBDIR, "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)] + [str(a) for a in args], check=True, capture_output=True, text=True,
                         timeout=120)
    return out.stdout.strip().splitlines()


def test_config1_chain_in_c(gpu, golden, tmp_path):
    lines = _build_and_run(tmp_path, "config1_chain")
    assert len(lines) == 3
    for line, name in zip(lines, ("G1", "G2", "G3")):
        exp = golden["cases"][name]["vu"]
        f = dict(kv.split("=") for kv in line.split())
        assert int(f["frames"]) == exp["frames"]
        assert int(f["peak"]) == exp["global_peak"]
        assert float(f["power"]) == exp["global_power"]


def test_batch_block_in_c(gpu, oracle, tmp_path):
    from oracle import oracle_ffi as of
    lines = _build_and_run(tmp_path, "batch_block")
    _, g = oracle.gain(2, 2, 1000, [750, 1250])
    for line, s in zip(lines, (0, 255)):
        want = oracle.gain_apply(g, oracle.chmap([1, 0], oracle.lcg(12345 + s, 2 * 4096), 2), 2)
        v = oracle.vu_new(2)
        oracle.vu_accumulate(v, want)
        _, r = oracle.vu_result(v)
        assert ("peak %d," % r.global_peak) in line
        assert ("power %.17g dB" % r.global_power) in line
        if s == 0:
            assert ("first frame %d %d," % (want[0], want[1])) in line


def test_group_server_in_c(gpu, golden, tmp_path):
    """A many-stream host loop on the operator API (group.h): 94 blocks of 512 frames of the 48 kHz
    sine at unity gain are golden vector G1 on every stream, however the blocks were pipelined."""
    lines = _build_and_run(tmp_path, "group_server", 8, 512, 92)
    exp = golden["cases"]["G1"]["vu"]
    assert lines[0].startswith("streams 8 block 512:")
    assert len(lines) == 3
    for line, s in zip(lines[1:], (0, 7)):
        assert line == "stream %d: frames %d peak %d power %.17g" % (
            s, exp["frames"], exp["global_peak"], exp["global_power"])
