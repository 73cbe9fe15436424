#!/usr/bin/env python3
"""Audit of the hand-scheduled tile kernels' generated code (make asm ASM_SRC=tile_launch_f32h / tile_launch_f32s in plssvm_amd/csrc,
then: python tests/tools/audit_hand_asm.py plssvm_amd/lib/asm/*gfx950.s).

The kernels tile_matvec_f32_s6h / _f3h keep their B fragments in v[224:255], registers that only the generated asm groups
(lssvm_s6w_groups.inc) may touch (the software-pipelined tile_matvec_f32_f3p: everything from v64 / a64, lssvm_f3p_tiles.inc), and their accumulators are written by MFMAs inside asm statements, where the compiler pads no hazards.
Three things can silently break that contract, none of which the compiler reports:
  1. compiler-generated code that touches v224 and above (the register cap not holding: round 3 found amdgpu_num_vgpr(224) ineffective on
     gfx950, the attribute counts half registers -- NaNs);
  2. compiler-generated code that reads or writes an accumulator within a few instructions behind the MFMA that produces it (register
     copies at a branch merge, spills, epilogue instructions hoisted between the groups): an XDL write needs wait states before a VALU
     access that nobody inserts.  Only the results of a group's last two MFMAs can still be in flight behind the group;
  3. scratch traffic inside a loop (a reload is a vector-memory operation whose vmcnt(0) drains the LDS-DMA queue: slow, not wrong).
Exit code 1 if 1. or 2. is found in any hand-scheduled kernel."""
import re
import subprocess
import sys

HAND = re.compile(r"tile_matvec_f32_(s6h|f3h|f3p|pair)")


def first_private(name):
    """first register of the kernel's private range (VGPR and AGPR alike): s6h / f3h keep v[224:255], the software-pipelined f3p everything from 64"""
    return 64 if "f3p" in name else 224
NEAR = 4  # compiler instructions behind an MFMA group inside which an accumulator access counts as too early


def regs_of(text):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def audit(path):
    txt = open(path).read()
    bad = 0
    names = []
    for f in re.split(r"\n(?=\s*\.globl\s)", txt):
        m = re.search(r"\.globl\s+(\S+)", f)
        if not m or not HAND.search(m.group(1)):
            continue
        name = m.group(1)
        inasm = had_mfma = False
        since = None
        last_dst = set()
        in_loop = False
        maxreg = -1
        trespass, early, scratch_loop, scratch_any = [], [], 0, 0
        for no, line in enumerate(f.split("\n")):
            t = line.strip()
            if re.match(r"^\.?LBB\d+_\d+:", t) or t.startswith("; %bb."):
                in_loop = "in Loop" in line or "Loop Header" in line
            if t.startswith(";;#ASMSTART"):
                inasm, had_mfma, order = True, False, []
                continue
            if t.startswith(";;#ASMEND"):
                inasm = False
                if had_mfma:
                    # only the results of the group's LAST two MFMAs can still be in flight behind it (16 cycles per MFMA, 4 passes)
                    since, last_dst = 0, set().union(*order[-2:])
                continue
            if inasm:
                mm = re.match(r"v_mfma_\S+\s+v\[(\d+):(\d+)\]", t)
                if mm:
                    had_mfma = True
                    order.append(set(range(int(mm.group(1)), int(mm.group(2)) + 1)))
                if t.startswith("s_nop") and since is not None:
                    since = None  # wait states inside a later asm statement (the epilogue's s_nop 15)
                continue
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            r = regs_of(t.split(";")[0])
            if r:
                maxreg = max(maxreg, max(r))
            if any(x >= first_private(name) for x in r) or any(int(m.group(1)) >= first_private(name) for m in re.finditer(r"\ba\[?(\d+)", t.split(";")[0])):
                trespass.append((no + 1, t[:90]))
            if "scratch_" in t:
                scratch_any += 1
                scratch_loop += 1 if in_loop else 0
            if since is not None:
                since += 1
                if since <= NEAR and (r & last_dst):
                    early.append((no + 1, since, t[:90]))
                if since > NEAR:
                    since = None
        names.append(name)
        status = "ok"
        if trespass or early:
            status = "BROKEN"
            bad += 1
        print(f"{status:6s} {name}: highest compiler VGPR v{maxreg}, scratch ops {scratch_any} ({scratch_loop} inside loops)")
        for no, t in trespass[:5]:
            print(f"        line {no}: compiler code touches the private registers (v{first_private(name)}+ / a{first_private(name)}+): {t}")
        for no, k, t in early[:5]:
            print(f"        line {no}: accumulator accessed {k} instruction(s) behind its MFMA group: {t}")
    return bad, names


def main():
    total, count = 0, 0
    for path in sys.argv[1:]:
        bad, names = audit(path)
        total += bad
        count += len(names)
    try:
        print(subprocess.run(["c++filt"], input="", capture_output=True, text=True).stdout, end="")
    except OSError:
        pass
    print(f"{count} hand-scheduled kernels audited, {total} broken")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
