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
  4. (every kernel, not only the hand-scheduled ones) a load into a register that a v_mfma_f64_16x16x4_f64 issued fewer than eight wait states
     earlier reads as its C operand: see audit_dgemm_srcc.
  5. (every kernel whose inline asm issues LDS-DMA) compiler code that defines or uses M0, which lds_dma16 writes without being able to declare it.
Exit code 1 if 1., 2., 4. or 5. is found."""
import re
import subprocess
import sys

HAND = re.compile(r"tile_matvec_f32_(s6h|f3h|g6h|f3p|pair)")


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


LOAD = re.compile(r"^(ds_read\w*|ds_load\w*|global_load\w*|buffer_load\w*|scratch_load\w*|flat_load\w*)\s+(v\[\d+:\d+\]|v\d+)")
DGEMM = re.compile(r"^v_mfma_f64_16x16x4\w*\s+(v\[\d+:\d+\]),\s*\S+,\s*\S+,\s*(v\[\d+:\d+\])")
DGEMM_SRCC_WAIT_STATES = 8


def audit_dgemm_srcc(path):
    """EVERY kernel of the file (compiler-scheduled code too): a load that writes a register which a v_mfma_f64_16x16x4_f64 issued fewer than
    eight wait states earlier reads as its C operand is flagged: on gfx950 the load into the LAST register pair of C corrupts the last rows of C when
    it follows the MFMA by 3, 4 or 5 wait states (59 % / 56 % / 25 % of the lanes wrong; none at 6: tests/tools/repro/dgemm_srcc_war.hip, whose four
    ds_read_b64 put the last pair three instructions behind the first) -- and ROCm 7.2's hazard recognizer has no rule for it.  It needs C != D with the C
    registers dead behind the MFMA, which the compiler produces when it folds a splat start value into the first MFMA of several accumulators
    (the wrong instantiation of tile_matvec_f64_wide, round 3).  Returns the number of such sequences."""
    lines = [ln.strip().split(";")[0].strip() for ln in open(path).read().split("\n")]
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if re.match(r"^\.?LBB\d+_\d+:$", ln)}
    found = 0

    def scan(start, srcc, budget, where, depth=0):
        nonlocal found
        i = start
        while i < len(lines) and budget > 0:
            t = lines[i]
            i += 1
            if not t or t.startswith(".") and t.endswith(":") or t.startswith(";") or t.startswith("."):
                continue
            m = LOAD.match(t)
            if m and regs_of(m.group(2)) & srcc:
                found += 1
                print(f"HAZARD {path}: line {where}: v_mfma_f64_16x16x4 reads C = v[{min(srcc)}:{max(srcc)}], line {i}: '{t}' writes it {DGEMM_SRCC_WAIT_STATES - budget} wait state(s) later")
                return
            if t.startswith("s_endpgm") or t.startswith("s_setpc"):
                return
            mb = re.match(r"^s_c?branch\w*\s+(\S+)", t)
            if mb and depth < 2 and mb.group(1) in labels:
                scan(labels[mb.group(1)], srcc, budget - 1, where, depth + 1)
                if t.startswith("s_branch"):
                    return
            budget -= (int(t.split()[1]) + 1) if t.startswith("s_nop") else 1

    for no, t in enumerate(lines):
        m = DGEMM.match(t)
        if m and regs_of(m.group(1)) != regs_of(m.group(2)):
            scan(no + 1, regs_of(m.group(2)), DGEMM_SRCC_WAIT_STATES, no + 1)
    return found


def audit_m0(path):
    """EVERY kernel that issues LDS-DMA from inline asm (lds_dma16 writes M0 inside the statement, and a reserved register cannot be declared as a
    clobber): compiler-generated code of such a kernel must neither define nor use M0.  Returns the number of kernels that do."""
    bad = 0
    for f in re.split(r"\n(?=\s*\.globl\s)", open(path).read()):
        m = re.search(r"\.globl\s+(\S+)", f)
        if not m:
            continue
        inasm, dma, touches = False, False, []
        for no, line in enumerate(f.split("\n")):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                inasm = True
            elif t.startswith(";;#ASMEND"):
                inasm = False
            elif inasm:
                dma = dma or "global_load_lds" in t
            elif t and t[0] not in ";." and re.search(r"\bm0\b", t.split(";")[0]):
                touches.append((no + 1, t[:90]))
        if dma and touches:
            bad += 1
            print(f"BROKEN {m.group(1)}: compiler code touches M0 in a kernel whose inline asm owns it: line {touches[0][0]}: {touches[0][1]}")
    return bad


def main():
    total, count = 0, 0
    for path in sys.argv[1:]:
        bad, names = audit(path)
        total += bad
        count += len(names)
        total += audit_dgemm_srcc(path)
        total += audit_m0(path)
    try:
        print(subprocess.run(["c++filt"], input="", capture_output=True, text=True).stdout, end="")
    except OSError:
        pass
    print(f"{count} hand-scheduled kernels audited (+ every kernel for loads into the C operand of an in-flight v_mfma_f64), {total} broken")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
