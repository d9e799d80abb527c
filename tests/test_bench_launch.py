"""bench.py starts its own rank processes for --gpus N > 1 (the driver runs `python bench.py
--gpus N` as one command).  Covered here on the CPU with COOLMIC_BENCH_DRYRUN=1: the launch
plumbing is real -- N fresh processes, RANK / WORLD_SIZE / MASTER_* handed down, gloo
rendezvous, config 5's node-id exchange over its own socket, the barrier and the
max-over-ranks clock, exactly one JSON line from rank 0 -- only the GPU work is left out."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=180):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.parametrize("n,workload", [(2, "c2"), (2, "c5"), (4, "c4")])
def test_self_launch_prints_one_line_from_rank0(n, workload):
    p = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--workload", workload],
             {"COOLMIC_BENCH_DRYRUN": "1"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["dry_run"] is True
    assert out["ranks_seen"] == list(range(n))
    assert out["clock_max_over_ranks"] == float(n)          # rank r reports 1 + r: the slowest counts
    assert out["node_id_same_on_all_ranks"] is True
    assert out["config"]["workload"] == workload
    # every rank left the warm-up loop after the same chunk (the decision is rank 0's, broadcast): ranks that
    # issue collectives in their steps issue the same number of them
    assert len(set(out["warmup_steps_all_ranks"])) == 1 and out["warmup_steps_all_ranks"][0] > 1
    # configs 4 / 5 at this N ride on the driver's one command: the keys of their legs
    for key in ("rccl_ranks", "ms_per_step_c4", "ms_per_step_c5", "matches_host_merge"):
        assert key in out["node_vu"], key


def test_single_rank_needs_no_launcher_and_no_torch():
    p = _run(["--gpus", "1", "--steps", "3", "--warmup", "1"], {"COOLMIC_BENCH_DRYRUN": "1"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads(p.stdout.decode().strip())
    assert out["n_gpus"] == 1 and "ranks_seen" not in out


def test_ranks_that_hang_are_ended_at_the_deadline():
    """every rank alive, one of them stuck for ever (a collective its peer never issued, a hung GPU): the
    launch has a wall-clock deadline, names the ranks it ends and fails -- it does not poll for ever"""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"],
             {"COOLMIC_BENCH_DRYRUN": "1", "COOLMIC_BENCH_DRYRUN_HANG_RANK": "1", "COOLMIC_BENCH_DEADLINE_S": "25"},
             timeout=120)
    assert p.returncode == 124, p.stderr.decode()[-2000:]
    assert p.stdout.decode().strip() == ""
    assert b"still running" in p.stderr and time.time() - t0 < 80


def test_a_hang_inside_the_multi_rank_legs_cannot_lose_the_line():
    """rank 1 never comes back from its config-4 / config-5 legs, rank 0 waits for it inside a collective: the
    watchdog around the legs lets rank 0 write its line -- with node_vu saying what happened -- and ends every
    rank; the launch succeeds, because value and roofline were measured before"""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"],
             {"COOLMIC_BENCH_DRYRUN": "1", "COOLMIC_BENCH_DRYRUN_LEGS_HANG_RANK": "1",
              "COOLMIC_BENCH_LEGS_TIMEOUT_S": "8"}, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "did not finish within 8 s" in out["node_vu"]["error"]
    assert time.time() - t0 < 60


def test_no_self_launch_under_a_profiler_preload():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"],
             {"COOLMIC_BENCH_DRYRUN": "1", "ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"}, timeout=60)
    assert p.returncode == 2 and b"Profile one rank directly" in p.stderr
    assert p.stdout.decode().strip() == ""


def test_a_failing_rank_fails_the_launch():
    """without a GPU (and no dry run) every rank exits with 'no HIP device': the parent must
    report failure and print no JSON line -- never hang at a barrier"""
    import __graft_entry__ as ge
    if ge.load_package().device_count() > 0:
        pytest.skip("needs a machine without a GPU")
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], timeout=120)
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""
    assert b"no HIP device" in p.stderr


def test_a_rank_that_dies_does_not_leave_the_others_at_the_rendezvous():
    """rank 1 exits before the rendezvous; rank 0 would wait for it for half an hour: the parent gives
    the survivors 20 s, ends them, reports the failure and prints no line"""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"],
             {"COOLMIC_BENCH_DRYRUN": "1", "COOLMIC_BENCH_DRYRUN_FAIL_RANK": "1"}, timeout=120)
    assert p.returncode == 3
    assert p.stdout.decode().strip() == ""
    assert time.time() - t0 < 90
