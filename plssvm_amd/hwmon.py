"""Board power and shader clock of a HIP device, read from the amdgpu hwmon files of its PCI function (an ordinary user can read them; no
counters, no profiler).  Used by bench.py (`roofline.board_power`: is the chip at its power cap while the timed steps run?) and by
tests/tools/power_trace.py.  Measurement aid only: nothing on the compute path depends on it, and every reader returns None where the
files are not there."""

from __future__ import annotations

import ctypes as C
import glob
import os
import threading
import time


def _read_number(path):
    try:
        with open(path) as f:
            return float(f.read().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def hwmon_of_hip_device(device: int = 0):
    """(hwmon directory, PCI bus id) of HIP device `device`, found through hipDeviceGetPCIBusId -- a box shows the cards of ALL its GPUs in /sys,
    only the ones this process was given run its kernels.  (None, bus id or None) if there is no such directory."""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, C.c_int(64), C.c_int(device)) != 0:
            return None, None
    except OSError:
        return None, None
    bus = buf.value.decode().lower()
    for cand in (bus, bus if bus.count(":") == 2 else "0000:" + bus):
        found = sorted(glob.glob(f"/sys/bus/pci/devices/{cand}/hwmon/hwmon*"))
        if found:
            return found[0], cand
    return None, bus


class PowerSampler(threading.Thread):
    """Samples (time, watts, shader GHz) of one device every `period` seconds until stop(); rows with a missing reading carry None."""

    def __init__(self, device: int = 0, period: float = 0.02):
        super().__init__(daemon=True)
        self.hwmon, self.bus = hwmon_of_hip_device(device)
        self.period, self.rows, self._stop_flag = period, [], False
        self.power_file = None
        if self.hwmon is not None:
            self.power_file = next((os.path.join(self.hwmon, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.hwmon, f))), None)
        self.clock_file = os.path.join(self.hwmon, "freq1_input") if self.hwmon is not None else None

    @property
    def available(self) -> bool:
        return self.power_file is not None

    def cap_watts(self):
        cap = _read_number(os.path.join(self.hwmon, "power1_cap")) if self.hwmon is not None else None
        return None if cap is None else cap * 1e-6

    def run(self):
        while not self._stop_flag:
            p = _read_number(self.power_file) if self.power_file else None
            f = _read_number(self.clock_file) if self.clock_file else None
            self.rows.append((time.time(), None if p is None else p * 1e-6, None if f is None else f * 1e-9))
            time.sleep(self.period)

    def stop(self):
        self._stop_flag = True
        if self.is_alive():
            self.join()

    def window(self, t0: float, t1: float, settle: float = 0.3):
        """(watts, GHz) lists of the samples in [t0 + settle (t1 - t0), t1]"""
        rows = [r for r in self.rows if t0 + settle * (t1 - t0) <= r[0] <= t1]
        return [r[1] for r in rows if r[1] is not None], [r[2] for r in rows if r[2] is not None]


def median(values):
    s = sorted(values)
    return None if not s else (s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2]))


def smi_snapshot(gpu_index: int = 0, timeout_s: float = 30.0):
    """What `amd-smi metric --json` reports for one GPU right now (an ordinary user may run it): the firmware's throttle-residency accumulators
    (`ppt_accumulated` counts the ticks of `accumulation_counter` during which the power limit held the clock down), the violation status flags, socket power,
    the energy counter and the shader clocks.  (wall time, dict) or None where the tool or the device is not there.  Takes about a second of HOST time: call it
    outside a timed region, or from a side thread."""
    import json
    import shutil
    import subprocess
    import sys

    # Under a profiler that pre-loads itself into every process (rocprofv3 --pmc: LD_PRELOAD / HSA_TOOLS_LIB) the tool's own start-up would initialise the GPU in
    # the helper process too, and amd-smi's "#!/usr/bin/env python3" is then an exec out of a GPU-initialised process, which this pool's boxes refuse (seen in
    # round 5's first PMC passes): no throttle reading in profiled runs.  Elsewhere the script is started by the interpreter directly -- no `env` hop.
    if any(k in os.environ for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    exe = shutil.which("amd-smi") or "/opt/rocm/bin/amd-smi"
    script = os.path.realpath(exe)
    try:
        with open(script, "rb") as f:
            first = f.readline()
        is_python = first.startswith(b"#!") and b"python" in first
    except OSError:
        return None
    cmd = [sys.executable, script] if is_python else [exe]
    data, t = None, time.time()
    # (first for this GPU alone -- on an 8-GPU node the full report takes seconds --, then, should the tool not know the flag, for all)
    for extra in (["-g", str(gpu_index)], []):
        try:
            t = time.time()
            out = subprocess.run(cmd + ["metric"] + extra + ["--json"], capture_output=True, text=True, timeout=timeout_s).stdout
            start = min((i for i in (out.find("{"), out.find("[")) if i >= 0), default=-1)
            data = json.loads(out[start:]) if start >= 0 else None
        except (OSError, ValueError, subprocess.SubprocessError):
            data = None
        if data:
            break
    if not data:
        return None
    gpus = data.get("gpu_data", data) if isinstance(data, dict) else data
    if not isinstance(gpus, list) or not gpus:
        return None
    g = next((x for x in gpus if isinstance(x, dict) and x.get("gpu") == gpu_index), gpus[0])

    def num(x):
        if isinstance(x, dict):
            x = x.get("value")
        return float(x) if isinstance(x, (int, float)) else None

    thr = g.get("throttle", {}) if isinstance(g.get("throttle"), dict) else {}
    clocks = [num(v.get("clk")) for k, v in sorted((g.get("clock") or {}).items()) if k.startswith("gfx_") and isinstance(v, dict)]
    snap = {"accumulation_counter": num(thr.get("accumulation_counter")),
            "residency": {k[:-len("_accumulated")]: num(thr.get(k)) for k in ("prochot_accumulated", "ppt_accumulated", "socket_thermal_accumulated", "vr_thermal_accumulated", "hbm_thermal_accumulated")},
            "status": {k[:-len("_violation_status")]: thr.get(k) for k in ("prochot_violation_status", "ppt_violation_status", "socket_thermal_violation_status", "vr_thermal_violation_status",
                                                                            "hbm_thermal_violation_status") if isinstance(thr.get(k), str)},
            "socket_power_w": num((g.get("power") or {}).get("socket_power")),
            "energy_j": num((g.get("energy") or {}).get("total_energy_consumption")),
            "gfx_clocks_mhz": [c for c in clocks if c is not None]}
    return t, snap


def throttle_between(s0, s1, t_begin: float, t_end: float):
    """Throttle residencies of the window [t_begin, t_end] (the timed region) from two smi_snapshot()s taken just outside it: the accumulators tick at a fixed
    rate (ticks per second = counter difference / wall difference of the snapshots), a throttler's residency is its tick difference in seconds -- reported as a
    fraction of the timed region, at most 1 (the chip idles between the snapshots and the region, where nothing throttles)."""
    if s0 is None or s1 is None:
        return None
    (w0, a), (w1, b) = s0, s1
    if a["accumulation_counter"] is None or b["accumulation_counter"] is None or w1 <= w0 or t_end <= t_begin:
        return None
    ticks = b["accumulation_counter"] - a["accumulation_counter"]
    if ticks <= 0:
        return None
    rate = ticks / (w1 - w0)
    out = {"ticks_per_s": rate, "snapshot_window_s": w1 - w0, "timed_region_s": t_end - t_begin, "residency_frac_of_timed_region": {}, "residency_s": {}}
    for name, v1 in b["residency"].items():
        v0 = a["residency"].get(name)
        if v0 is None or v1 is None:
            continue
        sec = (v1 - v0) / rate
        out["residency_s"][name] = sec
        out["residency_frac_of_timed_region"][name] = min(1.0, sec / (t_end - t_begin))
    if a["energy_j"] is not None and b["energy_j"] is not None:
        out["energy_j_between_snapshots"] = b["energy_j"] - a["energy_j"]
    return out
