#!/usr/bin/env python3
"""Generates gbrl_amd/csrc/predict_reg_asm.h: the inline-assembly text of the register-tile predict walk (predict_reg.hip).

Why generated: the walk is written against FIXED physical registers (the row tile lives in a VGPR bank the compiler never
allocates; record buffers and compare masks in named SGPRs), and one variant exists per (tree levels, padded outputs).  The
preprocessor cannot do the register arithmetic, so this script prints the strings; the header is committed, and
`python scripts/gen_predict_reg_asm.py --check` (tests/test_host.py) fails when the two drift apart.

Register map (see predict_reg.hip for the reasoning):
  v0 .. v87      the compiler's (amdgpu_num_vgpr(44): the cap is counted in VGPR+AGPR halves)
  v88 .. v215    the row tile: feature f of this lane's row in v[88 + f] (numeric), categorical ids behind the numeric features
  v216, v217     leaf offsets of the step's two trees
  v218 ..        leaf values of the step's two trees (DMAX floats each)
  s[26:27]       condition record pointer;  s28 ..  record buffers R0, R1 (one step = two trees), then the compare masks
"""
import argparse
import os
import sys

PREFETCH_STEPS = int(os.environ.get("GBRL_GEN_PREFETCH_STEPS", "0"))   # experiment: touch the records this many steps ahead (0 = off)
TB = 88            # first bank register
NF = 128           # bank size
LA, LB = 216, 217  # leaf registers
V0 = 218           # first value register


def vreg(r, n=1):
    return f"v{r}" if n == 1 else f"v[{r}:{r + n - 1}]"


def sreg(r, n=1):
    return f"s{r}" if n == 1 else f"s[{r}:{r + n - 1}]"


class Variant:
    def __init__(self, maxd, dmax):
        self.maxd, self.dmax = maxd, dmax
        self.dw = dmax // 4                      # floats per (slice, leaf)
        self.ls = 1 << maxd
        self.rec = self.ls * dmax * 4            # bytes of one tree's values
        self.slice = self.ls * self.dw * 4       # bytes of one slice
        self.tree_dw = 2 * maxd                  # condition dwords per tree
        self.step_dw = 2 * self.tree_dw
        self.step_bytes = self.step_dw * 4
        self.r0 = 28
        self.r1 = self.r0 + self.step_dw
        self.m0 = self.r1 + self.step_dw         # masks: 2 per level and tree
        self.s_end = self.m0 + 4 * maxd
        assert self.s_end <= 100, (maxd, self.s_end)
        self.va = V0
        self.vb = V0 + dmax
        self.v_end = V0 + 2 * dmax
        assert self.v_end <= 250
        self.sfx = f"D{maxd}O{dmax}"

    # --- pieces ---
    def loads(self, rbase, off):
        """s_load of one step's records into the buffer at rbase from s[26:27] + off"""
        out, dw, r, o = [], self.step_dw, rbase, off
        for width in (16, 8, 4):
            while dw >= width:
                out.append(f"s_load_dwordx{width} {sreg(r, width)}, s[26:27], {hex(o)}")
                dw -= width; r += width; o += 4 * width
        assert dw == 0
        return out

    def loads_one(self, rbase):
        out, dw, r, o = [], self.tree_dw, rbase, 0
        for width in (16, 8, 4):
            while dw >= width:
                out.append(f"s_load_dwordx{width} {sreg(r, width)}, s[26:27], {hex(o)}")
                dw -= width; r += width; o += 4 * width
        assert dw == 0
        return out

    def cmps(self, rbase, trees=2):
        """mode on, one (index, compare) per level and tree, mode off.  mask of (tree t, level d) = s[m0 + 2*(t*maxd + d)]"""
        out = []
        first = True
        for t in range(trees):
            for d in range(self.maxd):
                f = rbase + 2 * (t * self.maxd + d)
                m = self.m0 + 2 * (t * self.maxd + d)
                out.append(f"s_set_gpr_idx_on {sreg(f)}, gpr_idx(SRC0)" if first else f"s_set_gpr_idx_idx {sreg(f)}")
                first = False
                out.append(f"v_cmp_gt_f32_e64 {sreg(m, 2)}, {vreg(TB)}, {sreg(f + 1)}")
        out.append("s_set_gpr_idx_off")
        return out

    def addcs(self, trees=2):
        out = []
        regs = [LA, LB][:trees]
        for d in range(self.maxd):
            for t, l in enumerate(regs):
                m = self.m0 + 2 * (t * self.maxd + d)
                src = "0, 0" if d == 0 else f"{vreg(l)}, {vreg(l)}"
                out.append(f"v_addc_co_u32_e64 {vreg(l)}, vcc, {src}, {sreg(m, 2)}")
        return out

    def dsreads(self, trees=2):
        out = []
        sh = {1: 2, 2: 3, 4: 4, 8: 5}[self.dw]
        regs = [LA, LB][:trees]
        for l in regs:
            out.append(f"v_lshl_add_u32 {vreg(l)}, {vreg(l)}, {sh}, %[vb]")
        for t, l in enumerate(regs):
            base = self.va + t * self.dmax
            for sl in range(4):
                off = t * self.rec + sl * self.slice
                if self.dw == 1:
                    out.append(f"ds_read_b32 {vreg(base + sl)}, {vreg(l)} offset:{off}")
                elif self.dw == 2:
                    out.append(f"ds_read_b64 {vreg(base + 2 * sl, 2)}, {vreg(l)} offset:{off}")
                elif self.dw == 4:
                    out.append(f"ds_read_b128 {vreg(base + 4 * sl, 4)}, {vreg(l)} offset:{off}")
                else:
                    raise NotImplementedError
        out.append(f"s_add_u32 %[vb], %[vb], {trees * self.rec}")
        return out

    def fmas(self, trees=2):
        out = []
        for t in range(trees):
            base = self.va + t * self.dmax
            for pr in range(self.dmax // 2):
                out.append(f"v_pk_fma_f32 %[p{pr}], %[n{pr}], {vreg(base + 2 * pr, 2)}, %[p{pr}]")
        return out

    # --- whole statements ---
    def walk_steps(self):
        """`n` steps (>= 1) of two trees each, software pipelined: the values of step s are in flight while step s + 1 is searched;
        every s_waitcnt lgkmcnt(0) only meets requests that are a whole step old."""
        sb = self.step_bytes
        L = []
        L += ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"]
        L += self.loads(self.r0, 0) + self.loads(self.r1, sb)
        L += ["s_sub_u32 %[n], %[n], 1", "s_waitcnt lgkmcnt(0)"]
        L += self.cmps(self.r0) + self.loads(self.r0, 2 * sb) + self.addcs() + self.dsreads()
        L += ["s_cmp_eq_u32 %[n], 0", "s_cbranch_scc1 3f", "1:"]
        pf1 = [f"s_load_dword s100, s[26:27], {hex((3 + PREFETCH_STEPS) * sb)}"] if PREFETCH_STEPS else []
        pf0 = [f"s_load_dword s100, s[26:27], {hex((2 + PREFETCH_STEPS) * sb)}"] if PREFETCH_STEPS else []
        L += self.cmps(self.r1) + self.addcs() + ["s_waitcnt lgkmcnt(0)"] + self.loads(self.r1, 3 * sb) + pf1 + self.fmas() + self.dsreads()
        L += ["s_sub_u32 %[n], %[n], 1", "s_cmp_eq_u32 %[n], 0", "s_cbranch_scc1 3f"]
        L += self.cmps(self.r0) + self.addcs() + ["s_waitcnt lgkmcnt(0)", f"s_add_u32 s26, s26, {2 * sb}", "s_addc_u32 s27, s27, 0"]
        L += self.loads(self.r0, 2 * sb) + pf0 + self.fmas() + self.dsreads()
        L += ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b", "3:", "s_waitcnt lgkmcnt(0)"]
        L += self.fmas()
        return L

    def walk_one(self):
        L = ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"]
        L += self.loads_one(self.r0) + ["s_waitcnt lgkmcnt(0)"]
        L += self.cmps(self.r0, 1) + self.addcs(1) + self.dsreads(1) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(1)
        return L



# ======================================================================================================================
# Packed-code variant ("pc"): rows wider than the fp32 bank, or with categorical columns, are first rewritten as packed codes
# (engine_predict.hip / k_pack_codes): a numeric feature becomes a 16-bit field holding 0xffff - #{ensemble thresholds below x},
# a categorical column one INVERTED one-hot bit per category the ensemble mentions.  Every condition is then
#     D = word << shift          (field / bit to the top, zeros below)
#     mask = D < T               (numeric: T = (0xffff - rank) << 16;  categorical: T = 0x80000000, i.e. "the bit is clear")
# -- the same two instructions for both kinds, so mixed trees need no branch.  Register map:
#   v0 .. v61 compiler (amdgpu_num_vgpr(31));  v62 .. v221 the bank (160 words);  v222, v223 leaves / shifted words;
#   v224 .. v255 two sets of leaf values (the multiply-adds run two steps behind the search: the record buffer is single, so
#   every s_waitcnt drains BOTH counters -- value reads are issued right behind a wait and consumed two waits later).
#   s[26:27] record pointer;  s28 .. records of one step (3 dwords per level: word index, shift, T);  then the masks.
PC_TB = 62
PC_NF = 160
PC_LA, PC_LB = 222, 223
PC_V0 = 224


class VariantPC:
    def __init__(self, maxd, dmax):
        self.maxd, self.dmax = maxd, dmax
        self.dw = dmax // 4
        self.ls = 1 << maxd
        self.rec = self.ls * dmax * 4
        self.slice = self.ls * self.dw * 4
        self.tree_dw = 3 * maxd
        self.step_dw = 2 * self.tree_dw
        self.step_bytes = 4 * self.step_dw
        self.r0 = 28
        self.m0 = self.r0 + self.step_dw
        self.s_end = self.m0 + 4 * maxd
        assert self.s_end <= 100
        self.v_end = PC_V0 + 4 * dmax
        assert self.v_end <= 256
        self.sfx = f"D{maxd}O{dmax}"

    def loads(self, off, dw=None):
        out, dw, r, o = [], (self.step_dw if dw is None else dw), self.r0, off
        for width in (16, 8, 4, 2, 1):
            while dw >= width:
                out.append(f"s_load_dword{'x%d' % width if width > 1 else ''} {sreg(r, width)}, s[26:27], {hex(o)}")
                dw -= width; r += width; o += 4 * width
        assert dw == 0
        return out

    def cmps(self, trees=2):
        out, first = [], True
        for t in range(trees):
            for d in range(self.maxd):
                a = self.r0 + 3 * (t * self.maxd + d)
                m = self.m0 + 2 * (t * self.maxd + d)
                tmp = PC_LA if (d & 1) == 0 else PC_LB
                out.append(f"s_set_gpr_idx_on {sreg(a)}, gpr_idx(SRC1)" if first else f"s_set_gpr_idx_idx {sreg(a)}")
                first = False
                out.append(f"v_lshlrev_b32_e32 {vreg(tmp)}, {sreg(a + 1)}, {vreg(PC_TB)}")
                out.append(f"v_cmp_lt_u32_e64 {sreg(m, 2)}, {vreg(tmp)}, {sreg(a + 2)}")
        out.append("s_set_gpr_idx_off")
        return out

    def addcs(self, trees=2):
        out = []
        regs = [PC_LA, PC_LB][:trees]
        for d in range(self.maxd):
            for t, l in enumerate(regs):
                m = self.m0 + 2 * (t * self.maxd + d)
                src = "0, 0" if d == 0 else f"{vreg(l)}, {vreg(l)}"
                out.append(f"v_addc_co_u32_e64 {vreg(l)}, vcc, {src}, {sreg(m, 2)}")
        return out

    def dsreads(self, vset, trees=2):
        out = []
        sh = {1: 2, 2: 3}[self.dw]
        regs = [PC_LA, PC_LB][:trees]
        for l in regs:
            out.append(f"v_lshl_add_u32 {vreg(l)}, {vreg(l)}, {sh}, %[vb]")
        for t, l in enumerate(regs):
            base = PC_V0 + vset * 2 * self.dmax + t * self.dmax
            for sl in range(4):
                off = t * self.rec + sl * self.slice
                width = {1: "b32", 2: "b64"}[self.dw]
                out.append(f"ds_read_{width} {vreg(base + self.dw * sl, self.dw)}, {vreg(l)} offset:{off}")
        out.append(f"s_add_u32 %[vb], %[vb], {trees * self.rec}")
        return out

    def fmas(self, vset, trees=2):
        out = []
        for t in range(trees):
            base = PC_V0 + vset * 2 * self.dmax + t * self.dmax
            for pr in range(self.dmax // 2):
                out.append(f"v_pk_fma_f32 %[p{pr}], %[n{pr}], {vreg(base + 2 * pr, 2)}, %[p{pr}]")
        return out

    def walk_steps(self):
        """n >= 1 steps.  Iteration s: wait; value reads of step s - 1 -> set (s - 1) % 2; compares of s; records of s + 1; leaves of s;
        multiply-adds of step s - 2 from set s % 2."""
        sb = self.step_bytes
        adv = [f"s_add_u32 s26, s26, {sb}", "s_addc_u32 s27, s27, 0"]
        def it(s_par, with_ds, with_fma):
            L = ["s_waitcnt lgkmcnt(0)"]
            if with_ds:
                L += self.dsreads((s_par + 1) % 2)
            L += self.cmps() + adv + self.loads(0) + self.addcs()
            if with_fma:
                L += self.fmas(s_par)
            return L
        dec = ["s_sub_u32 %[n], %[n], 1", "s_cmp_eq_u32 %[n], 0"]
        L = ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"] + self.loads(0)
        L += it(0, False, False) + dec + ["s_cbranch_scc1 4f"]
        L += it(1, True, False) + dec + ["s_cbranch_scc1 5f"]
        L += ["1:"] + it(0, True, True) + dec + ["s_cbranch_scc1 6f"]
        L += it(1, True, True) + ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b"]
        # the last step had parity 1: its values go to set 1; pending multiply-adds: the step before it (set 0), then it
        L += ["5:"] + self.dsreads(1) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0) + self.fmas(1) + ["s_branch 9f"]
        L += ["6:"] + self.dsreads(0) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(1) + self.fmas(0) + ["s_branch 9f"]
        L += ["4:"] + self.dsreads(0) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0) + ["9:"]
        return L

    def walk_one(self):
        L = ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"] + self.loads(0, self.tree_dw) + ["s_waitcnt lgkmcnt(0)"]
        L += self.cmps(1) + self.addcs(1) + self.dsreads(0, 1) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0, 1)
        return L



# ======================================================================================================================
# Deep fp32 variant (round 5): 7 and 8 tree levels.  Two trees per step need 4 * maxd record dwords and 4 * maxd mask registers; with the
# records double-buffered (class Variant) 8 levels would need 124 scalar registers.  This variant keeps ONE record buffer and TWO value
# sets, with VariantPC's pipeline: every iteration waits once, issues the value reads of the previous step, the compares of this one, the
# record loads of the next (into the buffer the compares have just read), the leaf sums, and the multiply-adds of the step before last.
# Register map = class Variant's (bank v88 .. v215, leaves v216 / v217, values from v218: 4 * dmax registers).
class VariantDeep:
    def __init__(self, maxd, dmax):
        self.maxd, self.dmax = maxd, dmax
        self.dw = dmax // 4
        self.ls = 1 << maxd
        self.rec = self.ls * dmax * 4
        self.slice = self.ls * self.dw * 4
        self.tree_dw = 2 * maxd
        self.step_dw = 2 * self.tree_dw
        self.step_bytes = 4 * self.step_dw
        self.r0 = 28
        self.m0 = self.r0 + self.step_dw
        self.s_end = self.m0 + 4 * maxd
        assert self.s_end <= 100, (maxd, self.s_end)
        self.v_end = V0 + 4 * dmax
        assert self.v_end <= 256
        assert 2 * self.rec + 3 * self.slice < 65536      # ds_read offsets are 16 bits
        self.sfx = f"D{maxd}O{dmax}"

    def loads(self, off, dw=None):
        out, dw, r, o = [], (self.step_dw if dw is None else dw), self.r0, off
        for width in (16, 8, 4, 2, 1):
            while dw >= width:
                out.append(f"s_load_dword{'x%d' % width if width > 1 else ''} {sreg(r, width)}, s[26:27], {hex(o)}")
                dw -= width; r += width; o += 4 * width
        assert dw == 0
        return out

    def cmps(self, trees=2):
        out, first = [], True
        for t in range(trees):
            for d in range(self.maxd):
                f = self.r0 + 2 * (t * self.maxd + d)
                m = self.m0 + 2 * (t * self.maxd + d)
                out.append(f"s_set_gpr_idx_on {sreg(f)}, gpr_idx(SRC0)" if first else f"s_set_gpr_idx_idx {sreg(f)}")
                first = False
                out.append(f"v_cmp_gt_f32_e64 {sreg(m, 2)}, {vreg(TB)}, {sreg(f + 1)}")
        out.append("s_set_gpr_idx_off")
        return out

    def addcs(self, trees=2):
        out = []
        regs = [LA, LB][:trees]
        for d in range(self.maxd):
            for t, l in enumerate(regs):
                m = self.m0 + 2 * (t * self.maxd + d)
                src = "0, 0" if d == 0 else f"{vreg(l)}, {vreg(l)}"
                out.append(f"v_addc_co_u32_e64 {vreg(l)}, vcc, {src}, {sreg(m, 2)}")
        return out

    def dsreads(self, vset, trees=2):
        out = []
        sh = {1: 2, 2: 3}[self.dw]
        regs = [LA, LB][:trees]
        for l in regs:
            out.append(f"v_lshl_add_u32 {vreg(l)}, {vreg(l)}, {sh}, %[vb]")
        for t, l in enumerate(regs):
            base = V0 + vset * 2 * self.dmax + t * self.dmax
            for sl in range(4):
                off = t * self.rec + sl * self.slice
                width = {1: "b32", 2: "b64"}[self.dw]
                out.append(f"ds_read_{width} {vreg(base + self.dw * sl, self.dw)}, {vreg(l)} offset:{off}")
        out.append(f"s_add_u32 %[vb], %[vb], {trees * self.rec}")
        return out

    def fmas(self, vset, trees=2):
        out = []
        for t in range(trees):
            base = V0 + vset * 2 * self.dmax + t * self.dmax
            for pr in range(self.dmax // 2):
                out.append(f"v_pk_fma_f32 %[p{pr}], %[n{pr}], {vreg(base + 2 * pr, 2)}, %[p{pr}]")
        return out

    def walk_steps(self):
        """n >= 1 steps; the schedule of VariantPC.walk_steps."""
        sb = self.step_bytes
        adv = [f"s_add_u32 s26, s26, {sb}", "s_addc_u32 s27, s27, 0"]
        def it(s_par, with_ds, with_fma):
            L = ["s_waitcnt lgkmcnt(0)"]
            if with_ds:
                L += self.dsreads((s_par + 1) % 2)
            L += self.cmps() + adv + self.loads(0) + self.addcs()
            if with_fma:
                L += self.fmas(s_par)
            return L
        dec = ["s_sub_u32 %[n], %[n], 1", "s_cmp_eq_u32 %[n], 0"]
        L = ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"] + self.loads(0)
        L += it(0, False, False) + dec + ["s_cbranch_scc1 4f"]
        L += it(1, True, False) + dec + ["s_cbranch_scc1 5f"]
        L += ["1:"] + it(0, True, True) + dec + ["s_cbranch_scc1 6f"]
        L += it(1, True, True) + ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b"]
        L += ["5:"] + self.dsreads(1) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0) + self.fmas(1) + ["s_branch 9f"]
        L += ["6:"] + self.dsreads(0) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(1) + self.fmas(0) + ["s_branch 9f"]
        L += ["4:"] + self.dsreads(0) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0) + ["9:"]
        return L

    def walk_one(self):
        L = ["s_mov_b32 s26, %[cpl]", "s_mov_b32 s27, %[cph]"] + self.loads(0, self.tree_dw) + ["s_waitcnt lgkmcnt(0)"]
        L += self.cmps(1) + self.addcs(1) + self.dsreads(0, 1) + ["s_waitcnt lgkmcnt(0)"] + self.fmas(0, 1)
        return L


def load_tile_pc():
    L = ["s_getpc_b64 s[26:27]", "s_add_u32 s26, s26, %[skip]", "s_addc_u32 s27, s27, 0", "s_setpc_b64 s[26:27]"]
    for piece in range(PC_NF // 4 - 1, -1, -1):
        L.append(f"global_load_dwordx4 {vreg(PC_TB + 4 * piece, 4)}, %[row], off offset:{16 * piece}")
    return L


def load_tile():
    """The lane's row -> bank, 16 bytes per load, LAST piece first: a computed jump skips the pieces a narrower row does not have
    (every global_load is 8 bytes of code)."""
    L = ["s_getpc_b64 s[26:27]", "s_add_u32 s26, s26, %[skip]", "s_addc_u32 s27, s27, 0", "s_setpc_b64 s[26:27]"]
    # s_getpc returns the address of the instruction behind it; three 4-byte instructions follow before the first load, so
    # %[skip] = 12 + 8 * (32 - pieces), in an SGPR (a literal operand would make the s_add 8 bytes long)
    for piece in range(NF // 4 - 1, -1, -1):
        L.append(f"global_load_dwordx4 {vreg(TB + 4 * piece, 4)}, %[row], off offset:{16 * piece}")
    return L


def cstr(lines):
    return " \\\n".join(f'    "{l}\\n\\t"' for l in lines)


def clob(prefix, lo, hi):
    names = [f'"{prefix}{r}"' for r in range(lo, hi)]
    rows = [", ".join(names[i:i + 16]) for i in range(0, len(names), 16)]
    return ", \\\n    ".join(rows)


def generate():
    out = []
    out.append("// predict_reg_asm.h -- GENERATED by scripts/gen_predict_reg_asm.py; do not edit (tests/test_host.py checks it is current).")
    out.append("// Inline-assembly text of the register-tile predict walk; register map and reasoning: predict_reg.hip.")
    out.append("#pragma once")
    out.append(f"#define PR_TILE_BASE {TB}")
    out.append(f"#define PR_TILE_REGS {NF}")
    out.append(f"#define PR_COMPILER_VGPR_HALF {TB // 2}")
    out.append("#define PR_CLOB_TILE \\\n    " + clob("v", TB, TB + NF))
    out.append("#define PR_ASM_LOAD_TILE \\\n" + cstr(load_tile()))
    out.append('#define PR_CLOB_LOAD_TILE "s26", "s27", "scc"')
    for maxd, dmax in ((6, 8), (4, 8), (6, 4), (4, 4), (6, 16), (4, 16)):
        v = Variant(maxd, dmax)
        out.append(f"// ---- {maxd} levels, {dmax} padded outputs: {v.rec} bytes of values per tree, {v.step_bytes} bytes of records per step")
        out.append(f"#define PR_CLOB_TEMPS_{v.sfx} \\\n    " + clob("v", LA, v.v_end))
        out.append(f"#define PR_CLOB_SGPR_{v.sfx} \\\n    " + clob("s", 26, max(v.s_end, 101 if PREFETCH_STEPS else 0)))
        out.append(f"#define PR_ASM_WALK_STEPS_{v.sfx} \\\n" + cstr(v.walk_steps()))
        out.append(f"#define PR_ASM_WALK_ONE_{v.sfx} \\\n" + cstr(v.walk_one()))
    for maxd, dmax in ((8, 8), (8, 4)):
        v = VariantDeep(maxd, dmax)
        out.append(f"// ---- {maxd} levels (single record buffer, two value sets), {dmax} padded outputs: {v.rec} bytes of values per tree, {v.step_bytes} bytes of records per step")
        out.append(f"#define PR_CLOB_TEMPS_{v.sfx} \\\n    " + clob("v", LA, v.v_end))
        out.append(f"#define PR_CLOB_SGPR_{v.sfx} \\\n    " + clob("s", 26, v.s_end))
        out.append(f"#define PR_ASM_WALK_STEPS_{v.sfx} \\\n" + cstr(v.walk_steps()))
        out.append(f"#define PR_ASM_WALK_ONE_{v.sfx} \\\n" + cstr(v.walk_one()))
    out.append("// ==== packed-code variant (rows wider than the fp32 bank, categorical columns): see VariantPC in the generator")
    out.append(f"#define PC_TILE_BASE {PC_TB}")
    out.append(f"#define PC_TILE_REGS {PC_NF}")
    out.append(f"#define PC_COMPILER_VGPR_HALF {PC_TB // 2}")
    out.append("#define PC_CLOB_TILE \\\n    " + clob("v", PC_TB, PC_TB + PC_NF))
    out.append("#define PC_ASM_LOAD_TILE \\\n" + cstr(load_tile_pc()))
    for maxd, dmax in ((6, 8), (4, 8), (6, 4), (4, 4)):
        v = VariantPC(maxd, dmax)
        out.append(f"// ---- packed codes, {maxd} levels, {dmax} padded outputs: {v.step_bytes} bytes of records per step")
        out.append(f"#define PC_CLOB_TEMPS_{v.sfx} \\\n    " + clob("v", PC_LA, v.v_end))
        out.append(f"#define PC_CLOB_SGPR_{v.sfx} \\\n    " + clob("s", 26, v.s_end))
        out.append(f"#define PC_ASM_WALK_STEPS_{v.sfx} \\\n" + cstr(v.walk_steps()))
        out.append(f"#define PC_ASM_WALK_ONE_{v.sfx} \\\n" + cstr(v.walk_one()))
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=None, help="write the header somewhere else (experiments: GBRL_GEN_PREFETCH_STEPS=n)")
    args = ap.parse_args()
    path = args.out or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gbrl_amd", "csrc", "predict_reg_asm.h")
    text = generate()
    if args.check:
        cur = open(path).read() if os.path.exists(path) else ""
        if cur != text:
            print("predict_reg_asm.h is stale: run scripts/gen_predict_reg_asm.py")
            sys.exit(1)
        print("predict_reg_asm.h is current")
        return
    with open(path, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
