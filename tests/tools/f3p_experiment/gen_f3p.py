#!/usr/bin/env python3
"""Generates lssvm_f3p_tiles.inc: the SOFTWARE-PIPELINED form of the f16x3 tile kernel (tile_matvec_f32_f3p, lssvm_tile_f32_pipe.hip.hpp).

One wave per SIMD (512 registers per lane).  While the matrix cores multiply tile t into one accumulator set, the vector ALU evaluates
the epilogue of tile t - 1 (kernel function, row sums, mirrored column sums) from the OTHER accumulator set -- placed by this generator
instruction by instruction into the issue slots between the MFMAs (a v_mfma_f32_16x16x32_f16 holds the SIMD's vector issue for 8 of
its 16 cycles; MI355X_MICROARCH.md "vector-instruction ISSUE cost").  With two waves per SIMD (tile_matvec_f32_f3h) the epilogue of one
wave and the MFMAs of the other do not interleave: the kernel's time is the SUM of both.

The generated functions are inline-asm statements on PRIVATE registers, which the compiler never allocates (the kernel is compiled with
amdgpu_num_vgpr(80): v0 ... v79 and a0 ... a79 belong to the compiler; see LSSVM_HAND_VGPR_CAP for what that attribute counts):

    v[80:127]   epilogue state: d_j / e_j of four column blocks (rotating), two column sums (+ their odd-row halves), swap temporary, row sums (8), d_i (8), c_i (8)
    v[128:191]  accumulator set 0: acc(rb, cb) = 128 + 4 (2 cb + rb)          (2 row blocks x 8 column blocks of 16 x 16)
    v[192:255]  accumulator set 1
    a[96:...]   the row panel: A fragments [plane][k32 step][row block], 4 registers each (MFMA operands may be AGPRs)
    a[224:255]  B fragments, two buffers of 4 column blocks (ds_read_b128 writes AGPRs directly)

Everything that depends on the tile (LDS addresses of the records, source pointers of the LDS-DMA, ...) comes in as asm operands from
the compiler-generated code around the statements.  The arithmetic and its ORDER are those of s6w_body: results are bit-identical to the
two-waves-per-SIMD kernels.

    python gen_f3p.py            writes lssvm_f3p_tiles.inc beside this script
"""
import sys

# timing experiments (results wrong): python gen_f3p.py --no-fillers | --no-lds-fillers | --no-dma | --no-exp
NO_FILLERS = "--no-fillers" in sys.argv
NO_LDS_FILLERS = "--no-lds-fillers" in sys.argv
NO_DMA = "--no-dma" in sys.argv
NO_EXP = "--no-exp" in sys.argv
NO_BARRIER = "--no-barrier" in sys.argv
NO_COLS = "--no-cols" in sys.argv      # timing: row sums only
HALF_ROWS = "--half" in sys.argv        # timing: the epilogue of every other column block only
B_IN_VGPR = "--b-in-vgpr" in sys.argv   # B fragments in v[64:95] instead of a[224:255] (needs amdgpu_num_vgpr(64))

V_TMP = 80
R_DJ = [80, 81, 82, 83]  # d_j of a column block: requested two blocks ahead, rotating by block % 4 (8 blocks per tile: the rotation carries over)
R_EJ = [84, 85, 86, 87]  # e_j = 2^c_j
R_COL = [88, 89]         # column sum of a block (ping-pong: the LDS store of block cb is issued while block cb + 1 is evaluated)
R_T0 = 90
R_COLB = [91, 92]        # the odd rows' half of a block's column sum (s6w_body adds a lane's four rows as two packed pairs: even rows, odd rows, then their sum)
R_ROWP = 96          # 8: rowpart[4 rb + e]
R_DI = 104           # 8: d_i[4 rb + e]
R_CI = 112           # 8: c_i[4 rb + e]  (first C operand of the rbf chains)
ACC = [128, 192]
A_BASE = 96
B_BUF = [224, 240]
SLOT = 16384

COST_TRANS, COST_VALU = 8, 4


def breg(buf, c):
    if B_IN_VGPR:
        lo = 64 + 16 * buf + 4 * c
        return f"v[{lo}:{lo + 3}]"
    lo = B_BUF[buf] + 4 * c
    return f"a[{lo}:{lo + 3}]"


def acc(s, rb, cb, e=None):
    base = ACC[s] + 4 * (2 * cb + rb)
    return f"v{base + e}" if e is not None else f"v[{base}:{base + 3}]"


class Variant:
    def __init__(self, kt, nk64, sym):
        self.kt, self.nk64, self.sym = kt, nk64, sym
        self.rbf = kt in ("rbff",)
        self.pla = 3 if self.rbf else 2
        self.nkc = 2 * nk64
        assert 16 * self.pla * nk64 <= 128, "row panel does not fit a[96:223]"
        self.name = f"{kt}_k{nk64}_{'sym' if sym else 'full'}"

    def afrag(self, p, kk, rb):
        base = A_BASE + 4 * ((p * 2 * self.nk64 + kk) * 2 + rb)
        return f"a[{base}:{base + 3}]"

    def row_plane(self, plane, q):
        if self.pla == 3:
            return (2 if q == 0 else 1) if plane == 0 else 0
        return q

    def phase(self, s):
        """ring slot of the first step of a tile whose accumulator set is s (= tile index parity)"""
        return (self.nkc * s) & 3


# ---------------------------------------------------------------------------------------------------------------- the epilogue as a list of fillers
def epilogue_fillers(v, s_prev, cols, next_epilogue=True):
    """The epilogue of the tile held in accumulator set s_prev: (valu, lds).
    valu: [(text, cost, kind)] in program order -- placed between the MFMAs by cost.
    lds:  [(ready, text)] LDS operations, each to be issued at the first MFMA-group head (behind its s_waitcnt, so that the head's wait for
          the B fragments never waits for one of THESE) once `ready` vector instructions of `valu` have been emitted:
            - d_j (and e_j = 2^c_j) of column block cb + 2 are requested while block cb is evaluated: an LDS read issued at the head of MFMA
              group g has landed once the wait at the head of group g + 1 has passed, and a block spans at least one head (24 MFMAs against
              groups of 16 or 8).  Blocks 0 and 1 were requested by the previous statement (of the NEXT epilogue: next_epilogue);
            - the store of a block's column sum is issued while the next block is evaluated (two column-sum registers)."""
    valu, lds = [], []
    if NO_FILLERS:
        return valu, lds
    if NO_COLS:
        cols = False
    for cb in range(8):
        if HALF_ROWS and (cb & 1):
            continue
        start = len(valu)
        nxt = cb + 2
        if nxt < 8:
            lds.append((start, f"ds_read_b32 v{R_DJ[nxt % 4]}, %[dcr_prev] offset:{64 * nxt}"))
            if v.rbf and cols:
                lds.append((start, f"ds_read_b32 v{R_EJ[nxt % 4]}, %[dcr_prev] offset:{512 + 64 * nxt}"))
        elif next_epilogue:
            n0 = nxt - 8  # block 0 / 1 of the next epilogue: the record of the tile the matrix cores are working on
            lds.append((start, f"ds_read_b32 v{R_DJ[n0 % 4]}, %[dcr_cur] offset:{64 * n0}"))
            if v.rbf and v.sym:
                lds.append((start, f"ds_read_b32 v{R_EJ[n0 % 4]}, %[dcr_cur] offset:{512 + 64 * n0}"))
        elems = [(rb, e) for rb in range(2) for e in range(4)]
        if v.rbf and not NO_EXP:
            for rb, e in elems:
                valu.append((f"v_exp_f32_e32 {acc(s_prev, rb, cb, e)}, {acc(s_prev, rb, cb, e)}", COST_TRANS, "trans"))
        col = R_COL[cb & 1]
        colb = R_COLB[cb & 1]
        first = [True, True]
        for rb, e in elems:
            kv = acc(s_prev, rb, cb, e)
            valu.append((f"v_fmac_f32_e32 v{R_ROWP + 4 * rb + e}, {kv}, v{R_DJ[cb % 4]}", COST_VALU, "valu"))
            if cols:
                # the order of s6w_body's v_pk_fma_f32 pairs: even rows into one sum, odd rows into the other (a chain that starts from 0 starts with a product)
                valu.append((f"{'v_mul_f32_e32' if first[e & 1] else 'v_fmac_f32_e32'} v{colb if e & 1 else col}, {kv}, v{R_DI + 4 * rb + e}", COST_VALU, "valu"))
                first[e & 1] = False
        if cols:
            valu.append((f"v_add_f32_e32 v{col}, v{col}, v{colb}", COST_VALU, "valu"))
            # the four lane groups hold different rows of the same column: two butterfly steps (sum_with_lane_xor32, then _xor16), the column's
            # factor 2^c_j, one store per lane group (same address, same value)
            valu.append((f"v_mov_b32_e32 v{R_T0}, v{col}", COST_VALU, "valu"))
            valu.append((f"v_permlane32_swap_b32 v{col}, v{R_T0}", COST_VALU, "perm"))
            valu.append((f"v_add_f32_e32 v{col}, v{col}, v{R_T0}", COST_VALU, "valu"))
            valu.append((f"v_mov_b32_e32 v{R_T0}, v{col}", COST_VALU, "valu"))
            valu.append((f"v_permlane16_swap_b32 v{col}, v{R_T0}", COST_VALU, "perm"))
            valu.append((f"v_add_f32_e32 v{col}, v{col}, v{R_T0}", COST_VALU, "valu"))
            if v.rbf:
                valu.append((f"v_mul_f32_e32 v{col}, v{col}, v{R_EJ[cb % 4]}", COST_VALU, "valu"))
            lds.append((len(valu), f"ds_write_b32 %[cw], v{col} offset:{64 * cb}"))
    if NO_LDS_FILLERS:
        lds = []
    return valu, lds


def emit_filler(lines, f):
    text, _, kind = f
    if kind == "perm":
        # a VALU write needs two wait states before v_permlane*_swap reads the register (the instruction in front is the v_mov that wrote it)
        lines.append("s_nop 1")
    lines.append(text)


# ---------------------------------------------------------------------------------------------------------------- one tile of MFMAs
def tile_stream(v, s_cur, valu, lds, with_dma=True):
    """Returns the asm lines of the two statements of a tile whose MFMAs go to accumulator set s_cur: part A = the first two groups and the
    first hand-over (the compiler-side flush of the column sums follows it), part B = the rest."""
    ph = v.phase(s_cur)
    n_mfma = sum(8 * (2 - (kc % 2)) for kc in range(v.nkc)) * 4
    total_cost = sum(c for _, c, _ in valu)
    parts = {"A": [], "B": []}
    state = {"done": 0, "cost": 0, "fi": 0, "li": 0, "heads": 0}
    issued_at_head = {}   # index into lds -> head number at issue

    def fill(lines, upto_mfma):
        target = total_cost * upto_mfma / n_mfma
        while state["fi"] < len(valu) and state["cost"] + valu[state["fi"]][1] / 2 <= target:
            emit_filler(lines, valu[state["fi"]])
            state["cost"] += valu[state["fi"]][1]
            state["fi"] += 1

    def head_lds(lines):
        """behind a group head's wait: the LDS operations of the epilogue that have become ready"""
        state["heads"] += 1
        while state["li"] < len(lds) and lds[state["li"]][0] <= state["fi"]:
            lines.append(lds[state["li"]][1])
            issued_at_head[state["li"]] = state["heads"]
            state["li"] += 1

    for kc in range(v.nkc):
        plane, chunk = kc % 2, kc // 2
        slot = (ph + kc) & 3
        slot_next = (ph + kc + 1) & 3
        nq = 2 - plane
        for mm in range(4):
            kk, cbh, cur = mm >> 1, mm & 1, mm & 1
            lines = parts["A"] if (kc == 0 and mm < 2) else parts["B"]
            if mm == 2:
                # hand-over of chunk step + 1: this wave's DMA of it is complete once all but its 4 youngest DMA instructions are; the barrier
                # publishes every wave's part.  (kc == 0: the LDS writes of the previous epilogue's column sums must have completed as well.)
                if kc == 0:
                    la = parts["A"]
                    la.append(("s_waitcnt vmcnt(0) lgkmcnt(0)" if NO_DMA else "s_waitcnt vmcnt(4) lgkmcnt(0)") if v.sym else "s_waitcnt vmcnt(4)")
                    if not NO_BARRIER:
                        la.append("s_barrier")
                else:
                    lines.append("s_waitcnt vmcnt(0)" if NO_DMA else "s_waitcnt vmcnt(4)")
                    if not NO_BARRIER:
                        lines.append("s_barrier")
                # the record of the next tile travels right BEFORE that tile's first chunk (chunk step + 3 is the first chunk of tile t + 1 when kc + 3 == nkc)
                if with_dma and not NO_DMA and (kc + 3) % v.nkc == 0:
                    lines.append("s_mov_b64 exec, 0xffff")
                    lines.append("s_mov_b32 m0, %[dc_m0]")
                    lines.append("s_nop 0")
                    lines.append("global_load_lds_dwordx4 %[dc_off], %[dc_src]")
                    lines.append("s_mov_b64 exec, -1")
            # B fragments of the NEXT group into the other buffer
            if mm < 3:
                nkk, nh = (mm + 1) >> 1, (mm + 1) & 1
                addr, off0 = f"%[rd{nkk & 1}]", slot * SLOT + 4 * nh * 2048
            else:
                addr, off0 = "%[rd0]", slot_next * SLOT
            for c in range(4):
                lines.append(f"ds_read_b128 {breg(1 - cur, c)}, {addr} offset:{off0 + c * 2048}")
            lines.append("s_waitcnt lgkmcnt(4)")
            head_lds(lines)
            dma_i = 0
            for q in range(nq):
                for c in range(4):
                    cb = 4 * cbh + c
                    for rb in range(2):
                        a_op = v.afrag(v.row_plane(plane, q), 2 * chunk + kk, rb)
                        d = acc(s_cur, rb, cb)
                        if kc == 0 and kk == 0 and q == 0:
                            src_c = f"v[{R_CI + 4 * rb}:{R_CI + 4 * rb + 3}]" if v.rbf else "0"
                        else:
                            src_c = d
                        lines.append(f"v_mfma_f32_16x16x32_f16 {d}, {a_op}, {breg(cur, c)}, {src_c}")
                        state["done"] += 1
                        # the four LDS-DMA instructions of chunk step + 3 go behind the hand-over of this step, two per group
                        if with_dma and not NO_DMA and mm >= 2 and q == 0 and rb == 1 and c in (0, 2):
                            i = (mm - 2) * 2 + dma_i
                            dma_i += 1
                            dst_slot = (ph + kc + 3) & 3
                            lines.append(f"s_add_u32 m0, %[m0base], {dst_slot * SLOT + i * 1024}")
                            lines.append("s_nop 0")
                            lines.append(f"global_load_lds_dwordx4 %[dma{i}], %[src{kc}]")
                        fill(lines, state["done"])
    # whatever is left: vector instructions the cost rounding kept back, then the LDS operations that never met a head (the last block's store)
    while state["fi"] < len(valu):
        emit_filler(parts["B"], valu[state["fi"]])
        state["fi"] += 1
    state["heads"] += 1
    while state["li"] < len(lds):
        parts["B"].append(lds[state["li"]][1])
        issued_at_head[state["li"]] = state["heads"]
        state["li"] += 1
    return parts["A"], parts["B"]


def check_read_distance(lines):
    if NO_FILLERS or NO_LDS_FILLERS or NO_COLS or HALF_ROWS:
        return
    """every filler register loaded by ds_read_b32 must meet a group-head wait (s_waitcnt lgkmcnt) before its first use"""
    pending = {}
    for ln in lines:
        t = ln.split()
        if t[0] == "ds_read_b32":
            pending[t[1].rstrip(",")] = True
        elif t[0] == "s_waitcnt" and "lgkmcnt" in ln:
            pending.clear()
        else:
            for reg in list(pending):
                if (reg + ",") in ln + "," or ln.endswith(reg):
                    raise SystemExit(f"gen_f3p: {reg} is used before the wait that covers its ds_read: {ln}")


# ---------------------------------------------------------------------------------------------------------------- C++ wrappers
def cxx_asm(lines):
    return "\n        ".join('"' + ln + '\\n\\t"' for ln in lines)


def gen_variant(v):
    out = []
    ops_a = '"v"(rd0), "v"(rd1), "v"(dcr_prev), "v"(dcr_cur), "v"(cw)'
    ops_b = ops_a + ', "v"(dma0), "v"(dma1), "v"(dma2), "v"(dma3), ' + ", ".join(f'"s"(src{k})' for k in range(v.nkc)) \
        + ', "s"(m0base), "s"(dc_m0), "v"(dc_off), "s"(dc_src)'
    names_a = '[rd0] "v"(rd0), [rd1] "v"(rd1), [dcr_prev] "v"(dcr_prev), [dcr_cur] "v"(dcr_cur), [cw] "v"(cw)'
    names_b = names_a + ', [dma0] "v"(dma0), [dma1] "v"(dma1), [dma2] "v"(dma2), [dma3] "v"(dma3), ' + ", ".join(f'[src{k}] "s"(src{k})' for k in range(v.nkc)) \
        + ', [m0base] "s"(m0base), [dc_m0] "s"(dc_m0), [dc_off] "v"(dc_off), [dc_src] "s"(dc_src)'
    del ops_a, ops_b
    sig_a = "unsigned rd0, unsigned rd1, unsigned dcr_prev, unsigned dcr_cur, unsigned cw"
    sig_b = sig_a + ", unsigned dma0, unsigned dma1, unsigned dma2, unsigned dma3, " + ", ".join(f"const char *src{k}" for k in range(v.nkc)) \
        + ", unsigned m0base, unsigned dc_m0, unsigned dc_off, const char *dc_src"
    clob = '"memory", "scc", "v80", "v127", "v128", "v255", "a96", "a255"'
    for s_cur in (0, 1):
        for first in ((True, False) if s_cur == 0 else (False,)):
            if first:
                # first tile of a work item: no epilogue in flight; it requests d_j / e_j of column blocks 0 and 1 for the epilogue that follows
                lds = []
                for n0 in (0, 1):
                    lds.append((0, f"ds_read_b32 v{R_DJ[n0]}, %[dcr_cur] offset:{64 * n0}"))
                    if v.rbf and v.sym:
                        lds.append((0, f"ds_read_b32 v{R_EJ[n0]}, %[dcr_cur] offset:{512 + 64 * n0}"))
                pa, pb = tile_stream(v, s_cur, [], lds, True)
                tag = "first"
            else:
                valu, lds = epilogue_fillers(v, 1 - s_cur, v.sym)
                pa, pb = tile_stream(v, s_cur, valu, lds, True)
                tag = f"set{s_cur}"
            check_read_distance(pa + pb)
            out.append(f"/* {v.name}: MFMAs of a tile into accumulator set {s_cur}" + ("" if first else f", epilogue of the previous tile from set {1 - s_cur}") + " */")
            out.append(f"__device__ __forceinline__ void f3p_{v.name}_{tag}_a({sig_a}) {{\n    asm volatile(\n        {cxx_asm(pa)}\n        :\n        : {names_a}\n        : {clob});\n}}")
            out.append(f"__device__ __forceinline__ void f3p_{v.name}_{tag}_b({sig_b}) {{\n    asm volatile(\n        {cxx_asm(pb)}\n        :\n        : {names_b}\n        : {clob});\n}}")
    # drain: the epilogue of the last tile alone (its MFMAs have just been issued: the wait states an XDL write needs before a VALU read come
    # first).  Nothing to hide behind: every LDS request is waited for where it is needed.
    for s_prev in (0, 1):
        for cols in ((True, False) if v.sym else (False,)):
            valu, lds = epilogue_fillers(v, s_prev, cols, next_epilogue=False)
            lines = ["s_nop 15", "s_nop 3", "s_waitcnt lgkmcnt(0)"]
            li = 0
            for i, f in enumerate(valu):
                issued = False
                while li < len(lds) and lds[li][0] <= i:
                    lines.append(lds[li][1])
                    issued = issued or lds[li][1].startswith("ds_read")
                    li += 1
                if issued:
                    lines.append("s_waitcnt lgkmcnt(0)")
                emit_filler(lines, f)
            while li < len(lds):
                lines.append(lds[li][1])
                li += 1
            check_read_distance(lines)
            out.append(f"/* {v.name}: epilogue of the last tile of a work item, accumulator set {s_prev}, {'with' if cols else 'without'} the mirrored column sums */")
            out.append(f"__device__ __forceinline__ void f3p_{v.name}_drain{s_prev}_{'cols' if cols else 'rows'}(unsigned dcr_prev, unsigned dcr_cur, unsigned cw) {{\n"
                       f"    asm volatile(\n        {cxx_asm(lines)}\n        :\n        : [dcr_prev] \"v\"(dcr_prev), [dcr_cur] \"v\"(dcr_cur), [cw] \"v\"(cw)\n        : {clob});\n}}")
    return "\n\n".join(out)


def gen_common(v):
    """row panel into a[96:...], c_i / d_i / zeroed row sums into the private VGPRs, first B fragments, row sums back"""
    out = []
    loads = []
    for p in range(v.pla):
        for rb in range(2):
            for kk in range(2 * v.nk64):
                loads.append(f"global_load_dwordx4 {v.afrag(p, kk, rb)}, %[x{p}{rb}], off offset:{64 * kk}")
    ops = ", ".join(f'[x{p}{rb}] "v"(x{p}{rb})' for p in range(v.pla) for rb in range(2))
    sig = ", ".join(f"const void *x{p}{rb}" for p in range(v.pla) for rb in range(2))
    out.append(f"/* {v.name}: the row panel of this wave (planes x k32 steps x 2 row blocks) into the private AGPRs; waits for it */\n"
               f"__device__ __forceinline__ void f3p_{v.name}_load_panel({sig}) {{\n    asm volatile(\n        {cxx_asm(loads + ['s_waitcnt vmcnt(0)'])}\n        :\n        : {ops}\n        : \"memory\", \"a96\", \"a255\");\n}}")
    init = [f"ds_read_b128 v[{R_CI}:{R_CI + 3}], %[ci] offset:0", f"ds_read_b128 v[{R_CI + 4}:{R_CI + 7}], %[ci] offset:64",
            f"ds_read_b128 v[{R_DI}:{R_DI + 3}], %[di] offset:0", f"ds_read_b128 v[{R_DI + 4}:{R_DI + 7}], %[di] offset:64"]
    init += [f"v_mov_b32_e32 v{R_ROWP + i}, 0" for i in range(8)]
    # first B fragments (group 0 of step 0: k32 step 0, column half 0) of the work item's first tile (ring slot 0)
    init += [f"ds_read_b128 {breg(0, c)}, %[rd0] offset:{c * 2048}" for c in range(4)]
    out.append(f"/* c_i, d_i of the wave's rows (LDS: cis / dis + 128 wave + 16 g bytes), zero row sums, first B fragments */\n"
               f"__device__ __forceinline__ void f3p_init_state(unsigned ci, unsigned di, unsigned rd0) {{\n    asm volatile(\n        {cxx_asm(init)}\n        :\n"
               f"        : [ci] \"v\"(ci), [di] \"v\"(di), [rd0] \"v\"(rd0)\n        : \"memory\", \"v80\", \"v127\", \"a224\", \"a255\");\n}}")
    outs = ", ".join(f'"=v"(r{i})' for i in range(8))
    movs = [f"v_mov_b32_e32 %{i}, v{R_ROWP + i}" for i in range(8)]
    sig = ", ".join(f"float &r{i}" for i in range(8))
    out.append(f"__device__ __forceinline__ void f3p_get_rowsums({sig}) {{\n    asm volatile(\n        {cxx_asm(movs)}\n        : {outs}\n        :\n        : \"v80\", \"v127\");\n}}")
    return "\n\n".join(out)


def main():
    variants = [Variant("rbff", 2, True)]
    text = ["/* GENERATED by gen_f3p.py -- do not edit.  See that script for the register map and the scheduling rules. */", "#pragma once", ""]
    text.append(gen_common(variants[0]))
    for v in variants:
        text.append(gen_variant(v))
    target = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = target[0] if target else __file__.replace("gen_f3p.py", "lssvm_f3p_tiles.inc")
    open(path, "w").write("\n\n".join(text) + "\n")


if __name__ == "__main__":
    main()
