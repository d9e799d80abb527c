"""benchlib/: the host-side pieces of bench.py that can be held to account without a GPU -- the CPU baseline's
thread choice under a cgroup quota, the workloads' gain forms, the PMC summary lookup, the trace summary tool."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib import cpu, legs  # noqa: E402


def test_cpu_baseline_uses_the_cpus_of_the_quota(monkeypatch):
    """one thread per CPU the process may actually use: min(visible threads, ceil(quota)); the figure of one thread
    per VISIBLE hardware thread is reported beside it, not as `value`"""
    calls = []

    class Lib:
        def oracle_bench_block(self, threads, streams, channels, frames, cmap, scale, gain, seed, chk):
            calls.append((threads, streams, channels, frames, scale, list(gain)))
            return 1.0                                   # seconds: the rate is the sample count

        def oracle_bench_chain(self, frames, scale, gain, chk):
            return 1.0

    from oracle import oracle_ffi
    monkeypatch.setattr(oracle_ffi, "load", lambda: Lib())
    monkeypatch.setattr(cpu, "cpu_quota", lambda: 15.5)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    r = cpu.cpu_baseline("c2", 2, [750, 1250], True)
    assert r["cores"] == 16 and r["visible_hardware_threads"] == 256 and r["kind"] == "port"
    assert calls[0][0] == 16 and calls[0][1] == 16 * (calls[0][1] // 16)            # 16 threads, whole streams each
    assert calls[0][4] == 1000 and calls[0][5] == [750, 1250]
    assert calls[1][0] == 1                                                          # the one-thread figure
    assert calls[2][0] == 256 and "all_visible_threads_Msamples_s" in r              # ... and the oversubscribed one
    # about 10-30 s of CPU work in the main run at ~400 Msamples/s per thread
    samples = calls[0][1] * calls[0][2] * calls[0][3]
    assert 1.5e9 < samples < 6e9
    # no quota: every visible thread, and nothing beside it
    calls.clear()
    monkeypatch.setattr(cpu, "cpu_quota", lambda: None)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    r = cpu.cpu_baseline("c4", 1, None, False)
    assert r["cores"] == 8 and "all_visible_threads_Msamples_s" not in r
    assert calls[0][4] == 0                                                          # gain disabled: the reference's scale 0


def test_workload_gain_forms():
    assert legs.workload_gains("c2", 2) == ([750, 1250], True)
    assert legs.workload_gains("c2ro", 2, "below") == ([900, 800], True)
    assert legs.workload_gains("c2ro", 2, "off") == (None, True)
    assert legs.workload_gains("c4", 1) == ([900], False)
    assert legs.workload_gains("c3", 1, "off") == (None, False)
    assert set(legs.GAIN_FORMS) == {"general", "below", "off"}


def test_traffic_is_only_quoted_for_the_same_workload_shape_and_gain_form():
    for w in ("c2", "c2ro", "c3", "x6"):
        path = os.path.join(ROOT, "profiles", "pmc_%s.json" % w)
        if not os.path.exists(path):
            pytest.skip("no PMC summary for %s" % w)
        pmc = json.load(open(path))
        t, src = legs.traffic_from_profiles(w, pmc["streams"], pmc["frames"], pmc.get("gain_form", "general"))
        assert t == pmc["hbm_bytes_per_launch"] and "separate rocprofv3 --pmc passes" in src
        assert 0.98 < t / pmc["algorithmic_bytes_per_launch"] < 1.05                 # no wasted re-reads
        assert legs.traffic_from_profiles(w, pmc["streams"] + 1, pmc["frames"]) == (None, None)
        assert legs.traffic_from_profiles(w, pmc["streams"], pmc["frames"], "some other form") == (None, None)


def test_trace_summary_separates_the_timed_region_from_the_ramp(tmp_path):
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    rows, t = [], 1000
    for i in range(30):                                  # 10 slow warm-up launches, 20 timed ones, another kernel between
        dur = 400000 if i < 10 else 330000
        rows.append({"Kernel_Name": "void cmhip::k_run_fast<2, true, false, true, 4, 4>(cmhip::RunArgs)",
                     "Start_Timestamp": t, "End_Timestamp": t + dur})
        t += dur + 5000
        rows.append({"Kernel_Name": "void cmhip::k_vu_pack<2>(...)", "Start_Timestamp": t, "End_Timestamp": t + 9000})
        t += 10000
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        w.writerows(rows)
    line = {"steps": 20, "roofline": {"kernel": "k_run_fast", "kernel_avg_ms": 0.335, "frac": 0.8,
                                      "algorithmic_bytes_per_launch": 2147483648}}
    (tmp_path / "line.json").write_text(json.dumps(line))
    out = tmp_path / "sum.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_summary.py"), str(tmp_path / "prof"),
                    str(tmp_path / "line.json"), str(out)], check=True, capture_output=True)
    s = json.load(open(out))
    assert s["launches_in_trace"] == 30 and s["timed_region_launches"] == 20
    assert abs(s["avg_ms_timed_region"] - 0.330) < 1e-9 and abs(s["avg_ms_all_launches"] - (0.4 * 10 + 0.33 * 20) / 30) < 1e-9
    assert abs(s["line_over_trace"] - 0.335 / 0.330) < 1e-9
    assert abs(s["frac_of_8TBs_timed_region"] - 2147483648 / 0.330e-3 / 8e12) < 1e-9
