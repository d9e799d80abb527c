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
                    "-L", LIBDIR, "-lcoolmic-dsp-hip", "-lpthread", "-Wl,-rpath," + LIBDIR, "-o", str(exe)],
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
    exp = golden["cases"]["G1"]["vu"]
    for streams, threads in ((8, 1), (40, 4)):           # (the second: the pump's reads spread over four threads)
        lines = _build_and_run(tmp_path, "group_server", streams, 512, 92, threads)
        assert lines[0].startswith("streams %d block 512 pull threads %d:" % (streams, threads))
        assert len(lines) == 3
        for line, s in zip(lines[1:], (0, streams - 1)):
            assert line == "stream %d: frames %d peak %d power %.17g" % (
                s, exp["frames"], exp["global_peak"], exp["global_power"])


def test_group_server_over_the_gpus_of_the_box(gpu, golden, tmp_path):
    """examples/group_server.c with its fifth argument: one process, one coolmic_group_t per GPU made by
    coolmic_group_new_on(), capture stream s in the group of GPU s % N, one pump thread per GPU, and at the end
    the node-global VU over all groups -- the groups' records merged on the host and, through cmhip_node_*, by
    RCCL, which must agree.  gpus = 0 takes every GPU the box has (one on the driver's box: N > 1 in one process
    is unmeasured on hardware); the results do not depend on N: every stream is golden vector G1, the node
    sees streams x G1's frames at G1's level."""
    exp = golden["cases"]["G1"]["vu"]
    streams = 10
    lines = _build_and_run(tmp_path, "group_server", streams, 512, 92, 1, 0)
    lines = [ln for ln in lines if ln.startswith(("streams ", "stream ", "node:"))]     # (librccl announces itself)
    assert len(lines) == 4
    n = gpu.device_count()
    assert lines[0].startswith("streams %d block 512 pull threads 1:" % streams)
    assert lines[0].endswith("on %d GPU(s), stream s on GPU s %% %d, one pump thread each" % (n, n))
    for line, s in zip(lines[1:3], (0, streams - 1)):
        assert line == "stream %d: frames %d peak %d power %.17g" % (s, exp["frames"], exp["global_peak"], exp["global_power"])
    assert lines[3] == "node: frames %d peak %d power %.17g (%d GPU(s); host merge == RCCL exchange)" % (
        streams * exp["frames"], exp["global_peak"], exp["global_power"], n)


def test_node_vu_in_c(gpu, oracle, tmp_path):
    """config 5's step loop from plain C (examples/node_vu.c): streams sharded over every GPU the
    box has (one thread per GPU), node-global VU per block through cmhip_node_* (RCCL).  The
    result does not depend on the number of GPUs: checked against the oracle over all streams."""
    S, T, blocks = 48, 2048, 6
    lines = _build_and_run(tmp_path, "node_vu", S, T, blocks)
    # (librccl announces itself on stdout when a communicator is made: version, host, library path)
    lines = [ln for ln in lines if ln.startswith(("gpus ", "block "))]
    assert lines[0].startswith("gpus ") and lines[0].endswith("streams %d frames %d blocks %d" % (S, T, blocks))
    assert len(lines) == 1 + blocks
    _, g = oracle.gain(1, 1, 1000, [900])
    pcm = [oracle.gain_apply(g, oracle.lcg(12345 + s, T * blocks), 1).astype(np.int64) for s in range(S)]
    for k in range(blocks):
        blk = np.stack([p[k * T:(k + 1) * T] for p in pcm])          # [stream][frame]
        power = oracle.lib.oracle_power_db(int((blk ** 2).sum()), S * T)
        mag = np.abs(blk)
        top = int(mag.max())
        frames_of_top = [int(np.argmax(mag[s] == top)) if (mag[s] == top).any() else T for s in range(S)]
        s_first = min(range(S), key=lambda s: (frames_of_top[s], s))  # earliest frame, then lowest stream
        peak = int(blk[s_first, frames_of_top[s_first]])
        assert lines[1 + k] == "block %d: frames %d peak %d power %.17g" % (k, S * T, peak, power)


def test_product_chain_in_c(gpu, oracle, tmp_path):
    """The reference's live wiring up to the encoder in plain C (examples/product_chain.c): with and
    without the tee the meter reports the same last window (20 reads of 512 frames of the sine, at the
    same place of the stream), which is the oracle's."""
    pulls = 400
    lines = _build_and_run(tmp_path, "product_chain", pulls)
    assert len(lines) == 5 and lines[4].startswith("no meter gain on :")
    rc_s, sine = oracle.sine_table(48000)
    assert rc_s == 0 and len(sine) == 48
    for gain_on in (0, 1):
        direct, tee = lines[2 * gain_on], lines[2 * gain_on + 1]
        assert direct.startswith("direct gain %s:" % ("on " if gain_on else "off"))
        assert tee.startswith("tee    gain %s:" % ("on " if gain_on else "off"))
        assert direct.split("last window:")[1] == tee.split("last window:")[1]
        assert "frames 10240 " in tee
    if True:
        total = (200 + pulls) * 512
        x = np.tile(np.asarray(sine, dtype=np.int16), total // 48 + 2)[total - 10240: total]
        for gain_on in (0, 1):
            y = x
            if gain_on:
                _, g = oracle.gain(1, 1, 1000, [900])
                y = oracle.gain_apply(g, x, 1)
            v = oracle.vu_new(1)
            oracle.vu_accumulate(v, y)
            _, r = oracle.vu_result(v)
            assert lines[2 * gain_on + 1].endswith("frames 10240 peak %d power %.17g" % (r.global_peak, r.global_power))
