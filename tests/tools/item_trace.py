#!/usr/bin/env python3
"""Where a work item of the 256-row tile kernel spends its time OUTSIDE the tile loop, and what a CU does between two items (round 5).

Needs a measurement build (tests/tools/build_pair_variant.sh trace "" -DLSSVM_ITEM_TRACE): wave 0 of every work item stamps s_memrealtime (100 MHz) at its entry, when
its row panel has arrived, when it enters and leaves the tile loop and at its end, plus HW_ID / XCC_ID.  This script traces the implicit matvecs of a running CG loop (steady state: the last launch is what is read) and prints, per
phase, the mean over the items and the share of the launch, and the idle gaps of the CUs between consecutive items.

    PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_v_trace/libplssvm_amd.so [LSSVM_MI355_PAIR_QUEUE=0] python3 tests/tools/item_trace.py [points] [features] [kernel] [option=value ...]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # (first: the library then binds to the HIP runtime torch brought)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    kernel = sys.argv[3] if len(sys.argv) > 3 else "rbf"
    lib = C.CDLL(_capi.LIB_PATH)
    lib.lssvm_debug_set_item_trace.argtypes = [C.c_void_p]
    X, y = make_blobs_pm1(N, d, seed=42, dtype=np.float32)
    cap = 1 << 20
    buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda:0")
    opts = [a for a in sys.argv[4:] if "=" in a]  # library options, e.g. j_chunk_head=1 j_chunk_tiles=24
    for o in opts:
        _capi.set_option(o.split("=")[0], int(o.split("=")[1]))
    steps = 40 if N <= 200_000 else 6
    with backend.ResidentProblem(Parameter(kernel_type=kernel), X) as prob:
        # STEADY state: the trace stays on over `steps` CG iterations (every launch stamps the same slots; what is read is the last launch, at the clock the loop holds)
        prob.cg_begin(y, 1e-30)
        prob.cg_step(10 if N <= 200_000 else 2)
        assert lib.lssvm_debug_set_item_trace(C.c_void_p(buf.data_ptr())) == 0
        prob.cg_step(steps)
        prob.synchronize()
        assert lib.lssvm_debug_set_item_trace(C.c_void_p(0)) == 0
        info = prob.info()
    print(f"options {opts}: j_chunk_tiles {info.get('j_chunk_tiles')}, kernel ms per matvec {info.get('matvec_kernel_ms_total', 0) / max(info.get('matvec_timed', 1), 1):.4f}")
    t = buf.cpu().numpy().reshape(cap, 8).astype(np.uint64)
    used = t[:, 4] != 0
    t = np.concatenate([t[used], np.nonzero(used)[0][:, None].astype(np.uint64)], axis=1)[:, [0, 1, 2, 3, 4, 5, 6, 8]]  # (column 7: the item's index in the launch)
    dump = os.environ.get("ITEM_TRACE_DUMP")
    if dump:
        np.save(dump, t)
    tick = 0.01  # us per tick of the 100 MHz clock
    # several launches per matvec (row-block bands) stamp the same slots: keep the LAST launch -- the items that entered after every earlier entry had ended
    order = np.argsort(t[:, 0])
    t = t[order]
    ended = np.maximum.accumulate(t[:, 4].astype(np.float64))
    cut = np.nonzero(t[1:, 0].astype(np.float64) > ended[:-1] + 500)[0]  # (5 us: a kernel boundary; inside a launch some item is always running)
    if len(cut):
        t = t[cut[-1] + 1:]
        print(f"({len(cut) + 1} launches in the buffer: the last one kept)")
    n = len(t)
    t0, t1, t2, t3, t4 = (t[:, k].astype(np.float64) * tick for k in range(5))
    base = t0.min()
    t0, t1, t2, t3, t4 = t0 - base, t1 - base, t2 - base, t3 - base, t4 - base
    tiles = t[:, 6].astype(np.float64)
    hw = t[:, 5]
    cu_key = ((hw >> np.uint64(32)) << np.uint64(16)) | (((hw >> np.uint64(13)) & np.uint64(7)) << np.uint64(8)) | (((hw >> np.uint64(12)) & np.uint64(1)) << np.uint64(4)) | ((hw >> np.uint64(8)) & np.uint64(15))
    span = t4.max()
    print(f"{N} x {d} {kernel}: {n} traced work items ({info['tile_kernel_name'] if 'tile_kernel_name' in info else 'pair kernel'}), launch span {span:.1f} us, {len(np.unique(cu_key))} CUs seen, "
          f"tiles per item: mean {tiles.mean():.1f}, max {tiles.max():.0f}, sum {tiles.sum():.0f}")
    per_tile = (t3 - t2) / tiles
    print(f"tile loop: {np.median(per_tile):.3f} us per tile (median; 10 % {np.percentile(per_tile, 10):.3f}, 90 % {np.percentile(per_tile, 90):.3f})")
    unit = float(np.median(per_tile))
    for name, x in (("entry -> row panel in registers (incl. the first three chunks' DMA)", t1 - t0), ("row panel -> tile loop (derive plane, barrier, first B fragments)", t2 - t1),
                    ("tile loop", t3 - t2), ("tile loop left -> end (last flush, row sums, stores)", t4 - t3)):
        print(f"  {name:78s} mean {x.mean():8.2f} us = {x.mean() / unit:6.2f} tiles   (median {np.median(x):8.2f}, max {x.max():8.2f});  sum / (CUs x span) = {x.sum() / (256 * span):.4f}")
    # per CU: the items in time order, the gaps between them, the idle time at the end of the launch
    gaps, first, last_idle, busy = [], [], [], []
    for k in np.unique(cu_key):
        m = cu_key == k
        order = np.argsort(t0[m])
        a0, a4 = t0[m][order], t4[m][order]
        first.append(a0[0])
        gaps.extend(list(a0[1:] - a4[:-1]))
        last_idle.append(span - a4.max())
        busy.append(float((a4 - a0).sum()))
    gaps = np.array(gaps if gaps else [0.0])
    print(f"per CU: first item enters {np.mean(first):.2f} us after the launch's first (max {np.max(first):.2f}); gap between two items {gaps.mean():.2f} us mean (median {np.median(gaps):.2f}, "
          f"90 % {np.percentile(gaps, 90):.2f}, max {gaps.max():.2f}; {len(gaps) / len(first):.2f} gaps per CU) = {gaps.mean() / unit:.2f} tiles; "
          f"idle at the end {np.mean(last_idle):.1f} us mean (max {np.max(last_idle):.1f}) = {np.mean(last_idle) / span:.4f} of the span")
    # per XCD: the hardware deals workgroup i of a launch to XCD i % 8 -- with one workgroup per item that fixes every XCD's share of the work whatever its pace
    xcc = (hw >> np.uint64(32)).astype(np.int64)
    pos = t[:, 7].astype(np.int64)
    print(f"items that ran on the XCD of their list lane (position % 8): {np.mean((pos % 8) == xcc):.3f}")
    for x in np.unique(xcc):
        m = xcc == x
        ends = [t4[m & (cu_key == k)].max() for k in np.unique(cu_key[m])]
        print(f"  XCD {x}: {int(m.sum()):5d} items, {int(tiles[m].sum()):6d} tiles, {np.median(per_tile[m]):.3f} us per tile, its CUs end at {np.mean(ends):8.1f} us on average, the last at {np.max(ends):8.1f}")
    print(f"gaps of more than 5 us between two items of a CU: {int((gaps > 5).sum())} (sum {gaps[gaps > 5].sum():.0f} us)")
    short = tiles < 0.75 * tiles.max()
    if short.any() and (~short).any():
        print(f"items shorter than 3/4 of the longest: {int(short.sum())} of {n}, {np.median(per_tile[short]):.3f} us per tile (the others {np.median(per_tile[~short]):.3f}); "
              f"entered at {np.median(t0[short]) / span:.2f} of the span (median; the others {np.median(t0[~short]) / span:.2f})")
    print(f"shares of CUs x span: in items {np.sum(busy) / (len(first) * span):.4f}, gaps {gaps.sum() / (len(first) * span):.4f}, before the first item {np.mean(first) / span:.4f}, after the last {np.mean(last_idle) / span:.4f}")


if __name__ == "__main__":
    main()
