"""How a test harness puts the RCCL stand-in into a process (tests/tools/mp_rank.py, tests/tools/fake_rccl/local_multi.py, bench.py --rccl-stand-in):

    1. the HIP runtime the process will use is loaded FIRST -- PyTorch's own copy (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) where torch is
       installed: the stand-in and the product library both need "libamdhip64.so.7" and bind to whichever object with that SONAME is loaded already;
       without this step the stand-in would pull in /opt/rocm's copy, torch would add its own beside it, and the second HIP runtime of a process finds
       no device (seen on the first GPU run of round 5);
    2. the stand-in itself, RTLD_LOCAL (torch keeps calling its own RCCL): the product's later dlopen("librccl.so.1") returns it by SONAME.

Nothing here is product code; the product library never looks for the stand-in."""

import ctypes
import importlib.util
import os

DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librccl.so.1")


def load(path: str = DEFAULT):
    spec = importlib.util.find_spec("torch")  # (does not import torch)
    if spec is not None and spec.submodule_search_locations:
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.isfile(hip):
            ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL(os.path.abspath(path))
