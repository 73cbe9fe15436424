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
