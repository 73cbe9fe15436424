"""Tracker-style YAML of a run -- the counterpart of the reference's performance tracker
(src/plssvm/detail/performance_tracker.cpp:139-190 `save`, :36-38 generic entries, :40-58 the `parameter` entry).

The reference collects ``tracking_entry{category, name, value}`` objects while it runs and appends ONE YAML document per
program run to ``--performance_tracking <file>`` (or dumps it to stderr): a ``meta_data`` block, then one block per category with
two-space indented ``name: value`` lines.  The categories this path produces are the reference's own:

    parameter   kernel_type, degree, gamma, coef0, cost, real_type            (performance_tracker.cpp:40-58)
    backend     backend, target_platform (+ num_devices)                      (backends/HIP/csvm.hip.cpp:59-60)
    cg          iterations, max_iterations, residuum, target_residuum,
                avg_iteration_time, epsilon, total_runtime                    (backends/OpenMP/csvm.cpp:167-174, csvm.hpp:318-320)
    data_set_read / model_write    timings of the files either side of the solve
                total_time                                                     (src/main_train.cpp:57)
"""

from __future__ import annotations

import datetime
import sys
from collections import OrderedDict

__all__ = ["PerformanceTracker"]


def _fmt(value) -> str:
    if isinstance(value, bool):
        return "true" if value else "false"
    if isinstance(value, str):
        return f'"{value}"'  # string entries are quoted (performance_tracker.cpp:36-38)
    return str(value)


class PerformanceTracker:
    def __init__(self):
        self.entries: "OrderedDict[str, list[tuple[str, object]]]" = OrderedDict()

    def add(self, category: str, name: str, value) -> None:
        self.entries.setdefault(category, []).append((name, value))

    def add_parameter(self, params, real_type) -> None:
        p = params
        gamma = "#data_points" if getattr(p, "gamma", None) is None else p.gamma  # performance_tracker.cpp:52
        for name, value in (("kernel_type", str(p.kernel_type)), ("degree", p.degree), ("gamma", gamma), ("coef0", p.coef0), ("cost", p.cost),
                            ("real_type", "float" if str(real_type).endswith("32") else "double")):
            self.entries.setdefault("parameter", []).append((name, value if not isinstance(value, str) else _Raw(value)))

    def add_cg_info(self, info: dict) -> None:
        """The ``cg/*`` entries of one solve from ``lssvm_cg_info`` (as a dict) -- csvm.cpp:167-174, csvm.hpp:318-320."""
        self.add("cg", "iterations", int(info["iterations"]))
        self.add("cg", "max_iterations", int(info["max_iterations"]))
        self.add("cg", "residuum", float(info["residuum"]))
        self.add("cg", "target_residuum", float(info["target_residuum"]))
        self.add("cg", "avg_iteration_time", _Raw(f"{float(info['avg_iteration_ms']):.6g}ms"))
        self.add("cg", "epsilon", float(info["epsilon"]))
        if "total_runtime_ms" in info:
            self.add("cg", "total_runtime", _Raw(f"{float(info['total_runtime_ms']):.0f}ms"))

    def add_backend(self, num_devices: int) -> None:
        self.add("backend", "backend", _Raw("mi355"))
        self.add("backend", "target_platform", _Raw("gpu_amd"))
        self.add("backend", "num_devices", int(num_devices))

    def dumps(self) -> str:
        from . import _capi

        out = ["---", "meta_data:",
               f'  date:                    "{datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}"',
               '  PLSSVM_TARGET_PLATFORMS: "gpu_amd"',
               "  commit:                  unknown",
               f"  version:                 plssvm_amd ABI {_capi.ABI_VERSION}",
               ""]
        for category, items in self.entries.items():
            if category:
                out.append(f"{category}:")
            for name, value in items:
                out.append(f"{'  ' if category else ''}{name}: {value.text if isinstance(value, _Raw) else _fmt(value)}")
            out.append("")
        return "\n".join(out) + "\n"

    def save(self, filename: str | None = None) -> None:
        """performance_tracker::save (performance_tracker.cpp:139-151): append to the file, or dump to stderr."""
        text = self.dumps()
        if filename:
            with open(filename, "a") as f:
                f.write(text)
        else:
            sys.stderr.write("\n" + text)
        self.entries.clear()


class _Raw:
    """A value written without quotes (enumerators and durations are formatted, not quoted, by the reference)."""

    def __init__(self, text: str):
        self.text = text
