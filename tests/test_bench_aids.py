"""CPU: the measurement aids of bench.py -- plssvm_amd/hwmon.py, the reader behind `roofline.board_power` (absent files mean "not available", never an
error; a fake hwmon directory yields the readings in watts / GHz), and the stamp that ties profiles/hbm_traffic.json to the kernel sources."""

import time

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
