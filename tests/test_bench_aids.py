"""CPU: the measurement aids of bench.py -- plssvm_amd/hwmon.py, the reader behind `roofline.board_power` (absent files mean "not available", never an
error; a fake hwmon directory yields the readings in watts / GHz), and the stamp that ties profiles/hbm_traffic.json to the kernel sources."""

import os
import sys
import time

from conftest import ROOT
from plssvm_amd import hwmon


def test_median():
    assert hwmon.median([]) is None
    assert hwmon.median([3.0]) == 3.0
    assert hwmon.median([4.0, 1.0, 3.0]) == 3.0
    assert hwmon.median([4.0, 1.0, 3.0, 2.0]) == 2.5


def test_sampler_without_a_device_is_unavailable_not_an_error(monkeypatch):
    monkeypatch.setattr(hwmon, "hwmon_of_hip_device", lambda device=0: (None, None))
    s = hwmon.PowerSampler(0)
    assert not s.available and s.cap_watts() is None
    s.stop()  # (never started)
    assert s.window(0.0, 1.0) == ([], [])


def test_sampler_reads_a_hwmon_directory(tmp_path, monkeypatch):
    (tmp_path / "power1_input").write_text("1364000000\n")   # microwatts
    (tmp_path / "freq1_input").write_text("2000000000\n")    # hertz
    (tmp_path / "power1_cap").write_text("1400000000\n")
    monkeypatch.setattr(hwmon, "hwmon_of_hip_device", lambda device=0: (str(tmp_path), "0000:8e:00.0"))
    s = hwmon.PowerSampler(0, period=0.005)
    assert s.available and s.cap_watts() == 1400.0
    t0 = time.time()
    s.start()
    time.sleep(0.1)
    s.stop()
    watts, ghz = s.window(t0, time.time(), settle=0.0)
    assert len(watts) >= 5 and set(watts) == {1364.0} and set(ghz) == {2.0}


def test_traffic_stamp_ignores_comments_and_white_space_only():
    """bench.kernel_source_hash: the stamp that ties profiles/hbm_traffic.json to the kernel sources hashes the CODE -- a comment edit keeps it,
    a changed token does not."""
    import sys

    from conftest import ROOT

    sys.path.insert(0, ROOT)
    import bench

    a = 'int a = 1; // one\n/* two\n lines */ const char *s = "// kept /* too */";\n\tchar c = \'"\';'
    b = 'int a = 1;\nconst char *s = "// kept /* too */"; char c = \'"\';  // other words'
    assert bench.strip_comments(a) == bench.strip_comments(b) == 'int a = 1; const char *s = "// kept /* too */"; char c = \'"\';'
    assert bench.strip_comments(a) != bench.strip_comments(a.replace("a = 1", "a = 2"))
    assert len(bench.kernel_source_hash()) == 16 and bench.kernel_source_hash() != bench.kernel_source_hash(code_only=False)


def test_throttle_residency_from_amd_smi_reports(monkeypatch, tmp_path):
    """bench.py's `roofline.board_power.throttle` (round 5): plssvm_amd/hwmon.py reads the firmware's throttle-residency accumulators through `amd-smi metric --json` just
    outside the timed region and turns their differences into residencies of the region.  Parsed here from a report captured on an MI355X box (tests/golden/
    amd_smi_metric_sample.json: the fields the code reads, as the tool printed them), without a GPU: the per-GPU query comes first and falls back to the full report,
    the script is started by the interpreter (no `env` hop), nothing is started under a preloading profiler, and the arithmetic of two snapshots."""
    import json
    import subprocess

    from plssvm_amd import hwmon

    sample = open(os.path.join(ROOT, "tests", "golden", "amd_smi_metric_sample.json")).read()
    calls = []

    class Result:
        def __init__(self, out):
            self.stdout = out

    def fake_run(cmd, **kwargs):
        calls.append(cmd)
        return Result("" if "-g" in cmd else "some banner line\n" + sample)

    monkeypatch.setattr(subprocess, "run", fake_run)
    tool = tmp_path / "amd-smi"
    tool.write_text("#!/usr/bin/env python3\n# stands for /opt/rocm/libexec/amdsmi_cli/amdsmi_cli.py\n")
    monkeypatch.setattr(hwmon.os.path, "realpath", lambda p: str(tool))
    for var in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "LD_PRELOAD"):
        monkeypatch.delenv(var, raising=False)
    snap = hwmon.smi_snapshot(0)
    assert snap is not None and [c[-3:] if "-g" in c else c[-2:] for c in calls] == [["-g", "0", "--json"], ["metric", "--json"]]
    assert all(c[0] == sys.executable for c in calls)  # started by the interpreter, not through "#!/usr/bin/env python3"
    t, s = snap
    assert s["accumulation_counter"] == 142412249 and s["residency"]["ppt"] == 610188 and s["residency"]["hbm_thermal"] == 0
    assert s["status"]["ppt"] == "NOT ACTIVE" and s["socket_power_w"] == 255 and len(s["gfx_clocks_mhz"]) == 8 and s["energy_j"] > 0
    later = (t + 10.0, json.loads(json.dumps(s)))
    later[1]["accumulation_counter"] += 10000
    later[1]["residency"]["ppt"] += 2500
    later[1]["energy_j"] += 9000.0
    thr = hwmon.throttle_between(snap, later, t + 1.0, t + 9.0)
    assert abs(thr["ticks_per_s"] - 1000.0) < 1e-9 and abs(thr["residency_s"]["ppt"] - 2.5) < 1e-9
    assert abs(thr["residency_frac_of_timed_region"]["ppt"] - 2.5 / 8.0) < 1e-12 and thr["residency_frac_of_timed_region"]["prochot"] == 0.0
    assert hwmon.throttle_between(snap, None, 0.0, 1.0) is None
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    calls.clear()
    assert hwmon.smi_snapshot(0) is None and calls == []  # under a preloading profiler the tool is not started at all
