"""The hierarchical sampler's iteration, one lane per chain, in gfx950 assembly with every register placed by hand (see tools/gen_hier_isa.py
for the why and the overall layout) — one kernel per (number of experiments, points per experiment): configure(ne, shape) sets the layout,
Main emits the kernel.  Ne = 3 (any shape of up to 8 points per experiment): phf_hier3_advance (4 + 4 + 4), phf_hier3_advance_s<shape>.
Ne = 4: phf_hier4_advance_s<shape>, with a third state tier in device-memory scratch.  FusedMain: phf_hier_fused_advance, ONE persistent grid
with a body per kernel of HIER_KERNELS, one work queue for every launch group of a run.

Each phase below names the C it restates (pyhillfit_amd/csrc/phf_hierarchical.hip: hier_advance_body, PHF_LDL_COLUMN;
pyhillfit_amd/csrc/phf_hier_model.h: phf_hier_log_target_n / phf_hier_target_half / phf_hier_draws_k) — same fp64 operations, same
order per value, so that chains, states and moments are bit-identical to the hipcc kernels and to the twin (oracle/phf_oracle.c).
"""
import struct

import gfx950_asm as A
import phf_isa_math as M
from gfx950_asm import EXEC, VCC, Lit, Neg, Reg
from gen_hier_isa import ARG_BYTES, ARG_OFF, Gen

# ---- what lives where: set by configure(ne, shape) before a kernel is built (one kernel per number of experiments and point shape) ------
# LDS, per wavefront: [slot][64 lanes] doubles.  slots 0..D-1 the running mean, D..2D-1 the diagonal d, then the elements of L listed in
# LDS_L (row i, column k < i); every other element of L stays in VGPRs.
# ... and the wavefront's UNIFORM values, read with one address for all lanes (a broadcast): the pair's points, experiment by experiment
# [ln c (n, padded to even)][y (n, padded to even)], then the prior's loc[5], 1/scale[5], shape-1[5].  (Scalar loads would do — but they
# share lgkmcnt with the LDS and return out of order, so every table look-up behind one waits for it: ~11 exposed scalar-memory latencies
# per iteration, measured as 20 % of the wavefront's cycles in s_waitcnt.)
WAVE_BASE = 8064                                    # behind the workgroup's tables (M.TABLE_BYTES, padded)
GLB_FIRST_ROW, G_AHEAD, G_PRE = 9, 3, 3              # Ne = 4: rows 9..12 of L live in the scratch tier, fetched three columns ahead
TIERS = {4: (9, 3), 5: (4, 2), 6: (2, 1)}                      # Ne -> (first row of L in the scratch tier, columns fetched ahead); Ne = 5: all but rows 1..3

RESIDENT = ["L2E64", "NLN2HI64", "NLN2LO64", "KE0", "KE1", "K100", "KL0", "KL1", "KL2", "LN2HI", "LN2LO", "LN10", "ISQRT2",
            "M746", "P710", "P40", "P6", "QUARTER", "LOGADD", "MBITS"]


def shape_code(shape):
    """the code of a point shape (phf_hier_points.points_per_expt): PHF_HIER_SHAPE(per, last) of phf_hier_model.h — every experiment `per`
    points, the last one `last` if that differs — where the shape has that form, else the list form; None for a shape neither expresses"""
    per, last = shape[0], shape[-1]
    if all(n == per for n in shape[:-1]) and 0 < per < 16 and 0 < last < 16:
        return per if last == per else per | (last << 4)
    if len(shape) <= 7 and all(0 < n < 16 for n in shape):         # the list form (ABI 7): bit 30 + a nibble per experiment
        return (1 << 30) | sum(n << (4 * i) for i, n in enumerate(shape))
    return None


def kernel_name(ne, shape):
    if ne == 3 and tuple(shape) == (4, 4, 4):
        return "phf_hier3_advance"
    return "phf_hier%d_advance_s%s" % (ne, "".join(str(n) for n in shape))


def configure(ne, shape):
    """the module's layout constants for a kernel of `ne` experiments with shape[i] points in experiment i"""
    global NE, D, TRI, S_TH, S_LT, S_MEAN, S_TRI, S_LOGA, S_NACC, SHAPE, N_PTS, PT_START, PT_OFF, PT_YOFF, LDS_L, SLOT_MEAN, SLOT_D, SLOT_L, NSLOTS
    global U_POINTS, U_LOC, U_ISC, U_SM1, U_BYTES, WAVE_LDS, LDS_BYTES, NAME, GLB_L, GLB_FIRST_ROW, G_AHEAD, G_STREAM
    G_STREAM = 0
    assert len(shape) == ne and all(1 <= n <= 8 for n in shape)
    NE, D = ne, 5 + 2 * ne
    TRI = D * (D + 1) // 2
    S_TH, S_LT, S_MEAN, S_TRI, S_LOGA, S_NACC = 0, D, D + 1, 2 * D + 1, 2 * D + 1 + TRI, 2 * D + 2 + TRI       # state rows (phf_hierarchical.hip)
    SHAPE, N_PTS, NAME = tuple(shape), sum(shape), kernel_name(ne, shape)
    PT_START = [sum(shape[:i]) for i in range(ne)]                     # experiment i's first point in the pair's arrays
    pad = [n + (n & 1) for n in shape]
    PT_OFF = [16 * sum(pad[:i]) for i in range(ne)]                    # its block in the uniform area: [ln c (pad)][y (pad)]
    PT_YOFF = [8 * pad[i] for i in range(ne)]
    if ne == 3:
        LDS_L = [(i, 0) for i in range(1, D)] + [(i, 1) for i in range(2, 5)]
        GLB_L = []
    elif ne == 5:
        # Ne = 5: 171 doubles of state per chain.  theta, the proposal, w and u alone are 60 doubles during the sweep: L lives in the scratch
        # tier but for rows 1..3 (one element in a register, five in the LDS slots the mean and d leave), streamed column by column
        GLB_FIRST_ROW, G_AHEAD = TIERS[5]
        GLB_L = [(i, kc) for kc in range(D) for i in range(max(kc + 1, GLB_FIRST_ROW), D)]
        LDS_L = [(2, 0), (3, 0), (2, 1), (3, 1), (3, 2)]
    elif ne == 6:
        # Ne = 6: 208 doubles of state per chain; theta, the proposal, w and u are 68 of them, the mean and d fill 34 of the 35 LDS slots (L[1][0]
        # the last): rows 2..16 of L go through the scratch tier ELEMENT BY ELEMENT — a window of G_STREAM loads in flight over the tier's
        # column-major order, an element's two updates (w_i, then L_ik) and its term of y_i once the column's beta is known — because a
        # whole column in registers (15 doubles, and the next one's on top) does not fit
        GLB_FIRST_ROW, G_AHEAD = TIERS[6]
        GLB_L = [(i, kc) for kc in range(D) for i in range(max(kc + 1, GLB_FIRST_ROW), D)]
        LDS_L = [(1, 0)]
        G_STREAM = 16
    else:
        GLB_FIRST_ROW, G_AHEAD = TIERS[4]
        # Ne = 4: 135 doubles of state per chain against 128 registers + 35 LDS slots per lane at two wavefronts per SIMD: a THIRD tier,
        # a scratch area in device memory (L2-resident: 21 KB per resident wavefront), holds the last rows of L — four elements per
        # column, loaded G_AHEAD columns ahead of their use and stored back as the sweep produces them
        GLB_L = [(i, kc) for kc in range(D) for i in range(max(kc + 1, GLB_FIRST_ROW), D)]          # column by column: the order of use
        # ... and of the rows above them the LATE columns' elements go to LDS (columns 4..6: by then the sweep has returned the registers
        # of w_0..w_3 and u_0..u_3; column 0 is where every register is taken)
        LDS_L = [(i, kc) for kc, first in ((4, 5), (5, 6), (6, 7)) for i in range(first, GLB_FIRST_ROW)]
    SLOT_MEAN, SLOT_D, SLOT_L = 0, D, 2 * D
    NSLOTS = 2 * D + len(LDS_L)
    U_POINTS, U_LOC = 0, 16 * sum(pad)
    U_ISC, U_SM1, U_BYTES = U_LOC + 40, U_LOC + 80, U_LOC + 128
    WAVE_LDS = NSLOTS * 512 + U_BYTES
    LDS_BYTES = WAVE_BASE + 4 * WAVE_LDS
    assert 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU"


configure(3, (4, 4, 4))


def dbl_words(x):
    lo, hi = struct.unpack("<II", struct.pack("<d", float(x)))
    return lo, hi


def tri_index(i, k):
    return i * (i + 1) // 2 + k


class Main(object):
    # how a block's state becomes visible to the wavefront that runs its next quantum (possibly on another XCD, whose L2 is not coherent with
    # this one's).  False: the sequence hipcc emits for an agent-scope release — buffer_wbl2 writes EVERY dirty line of this XCD's L2 back,
    # the scratch tier's among them: 52 of a C4 step's 61 GB of HBM traffic were those.  True: the state's stores and the moments' atomics are
    # system-scope (written through / performed at the memory side), nothing else needs to leave the L2
    WRITE_THROUGH_RELEASE = True

    def __init__(self):
        self.write_through_release = self.WRITE_THROUGH_RELEASE
        self.g = Gen(NAME, NSLOTS)
        self.k = self.g.k
        self.m = self.g.m
        self.c = self.g.c
        self.info = {}
        self.ARG_OFF = ARG_OFF

    # ------------------------------------------------------------------------------------------------------------ small helpers
    def arg(self, dst, name, extra=0):
        self.k.s_load(dst, self.g.kernarg, self.ARG_OFF[name] + extra)
        return dst

    def add64(self, dst, a, b):
        """dst (SGPR pair) = a + b, 64 bits; b a pair or a single (zero-extended)"""
        k = self.k
        k.sop("s_add_u32", dst.lo(), a.lo(), b.lo() if b.n == 2 else b)
        k.sop("s_addc_u32", dst.hi(), a.hi(), b.hi() if b.n == 2 else 0)

    def udiv(self, q, r, n, d, magic, tmp):
        """q = n / d, r = n % d for 32-bit unsigned n, d >= 1 with magic = min(floor(2^32 / d), 2^32 - 1) from the host: the estimate
        floor(n magic / 2^32) is q or q - 1 (n magic / 2^32 > n / d - n / 2^32 > n / d - 1), one correction step makes it exact"""
        k = self.k
        k.sop("s_mul_hi_u32", q, n, magic)
        k.sop("s_mul_i32", tmp, q, d)
        k.sop("s_sub_u32", r, n, tmp)
        k.sop("s_cmp_ge_u32", None, r, d)
        k.sop("s_cselect_b32", tmp, d, 0)
        k.sop("s_sub_u32", r, r, tmp)                        # (SCC is the borrow now)
        k.sop("s_cmp_lg_u32", None, tmp, 0)                  # corrected (d >= 1)?
        k.sop("s_addc_u32", q, q, 0)

    def lds_addr(self, slot):
        return self.v_lds, slot * 512

    def lds_load(self, dst, slot):
        self.k.ds_read(dst, self.v_lds, slot * 512)

    def lds_store(self, slot, src):
        self.k.ds_write(self.v_lds, src, slot * 512)

    def uload(self, dst, off):
        """dst (a pair, or a quad for two consecutive doubles) <- the wavefront's uniform area at byte offset `off` (same address in every lane)"""
        self.k.ds_read(dst, self.v_ulds, off)
        return dst

    def l_slot(self, i, kcol):
        return SLOT_L + LDS_L.index((i, kcol))

    def g_addr(self, key):
        """(VGPR offset, immediate) of scratch slot GLB_L.index(key) against the one SGPR base (slot 8): the lane's offset + 8 KB per group of
        16 slots in a VGPR of its own (scalar registers are the scarcer kind here), the slot within the group in the +-4 KB immediate"""
        gidx = GLB_L.index(key)
        return self.v_goff[gidx // 16], (gidx % 16 - 8) * 512

    def g_load(self, dst, key):
        voff, imm = self.g_addr(key)
        self.k.gload(dst, voff, self.s_wsb, imm)

    def g_store(self, key, src):
        voff, imm = self.g_addr(key)
        self.k.gstore(voff, src, self.s_wsb, imm)

    # ------------------------------------------------------------------------------------------------------------ prologue
    def ptr(self, name):
        r = self.k.sd()
        self.k.s_load(r, self.g.kernarg, self.ARG_OFF[name])
        return r

    def prologue_once(self):
        """what a wavefront does once: constants, the workgroup's tables, its own LDS areas, the registers that live as long as it does"""
        self.prologue_common()
        self.alloc_state()

    def prologue_common(self):
        k, g = self.k, self.g
        k.comment("---- once per wavefront: constants, tables -> LDS, LDS bases ----")
        assert self.ARG_OFF["total_waves"] == self.ARG_OFF["num_problems"] + 12 and self.ARG_OFF["chains"] == self.ARG_OFF["adapt_start"] + 12
        s_consts = self.ptr("consts")
        g.load_constants(s_consts, RESIDENT)
        # every scalar that lives as long as the wavefront, allocated in one block (temporaries come and go behind it: no holes)
        self.s_pid, self.s_g0, self.s_flags, self.s_c8 = k.s1(), k.s1(), k.s1(), k.s1()
        self.s_t, self.s_tend, self.s_first, self.s_adapt = k.s1(), k.s1(), k.s1(), k.s1()
        self.s_until, self.s_thin, self.s_mafter = k.s1(), k.s1(), k.s1()
        self.s_seedlo, self.s_seedhi = k.s1(), k.s1()
        self.s_wave, self.s_block, self.s_quant = k.s1(), k.s1(), k.s1()      # wave in the workgroup; the task: block, quantum
        self.s_nch8, self.s_rows, self.s_rstride, self.s_mom = k.sd(), k.sd(), k.sd(), k.sd()
        self.s_gamma, self.s_acc, self.s_queue = k.sd(), k.sd(), k.sd()
        self.s_wsb = k.sd() if GLB_L else None
        g.stage_tables(s_consts)
        k.free(s_consts)
        self.v_lane, self.v_lane8, self.v_lds, self.v_ulds, self.v_cid = k.v1(), k.v1(), k.v1(), k.v1(), k.v1()
        self.v_zero, self.v_infhi = k.v1(), k.v1()
        k.mov32(self.v_zero, 0)
        k.mov32(self.v_infhi, Lit(0x7ff00000))
        vt, tmp = k.v1(), k.s1()
        k.vop("v_lshrrev_b32_e32", vt, 6, g.tid)
        k.readfirstlane(self.s_wave, vt)
        k.vop("v_and_b32_e32", self.v_lane, 63, g.tid)
        k.vop("v_lshlrev_b32_e32", self.v_lane8, 3, self.v_lane)
        k.sop("s_mul_i32", tmp, self.s_wave, Lit(WAVE_LDS))
        k.sop("s_add_u32", tmp, tmp, Lit(WAVE_BASE))
        k.vop("v_add_u32_e32", self.v_lds, tmp, self.v_lane8)
        k.sop("s_add_u32", tmp, tmp, Lit(NSLOTS * 512))
        k.mov32(self.v_ulds, tmp)
        k.s_load(self.s_queue, g.kernarg, self.ARG_OFF["queue"])
        if GLB_L:
            # this wavefront's scratch: scratch + (workgroup * 4 + wave) * slots * 512 bytes; base j = slot 16 j + 8
            per_wave = len(GLB_L) * 512
            t2 = k.sd()
            k.s_load(self.s_wsb, g.kernarg, self.ARG_OFF["scratch"])
            k.sop("s_lshl_b32", tmp, g.wg_id, 2)
            k.sop("s_add_u32", tmp, tmp, self.s_wave)
            k.sop("s_mul_hi_u32", t2.hi(), tmp, Lit(per_wave))
            k.sop("s_mul_i32", t2.lo(), tmp, Lit(per_wave))
            k.sop("s_add_u32", t2.lo(), t2.lo(), Lit(8 * 512))
            k.sop("s_addc_u32", t2.hi(), t2.hi(), 0)
            self.add64(self.s_wsb, self.s_wsb, t2)
            k.free(t2)
        k.free(vt, tmp)

    def alloc_state(self):
        """the chain state's registers"""
        k = self.k
        self.th = [k.vd() for _ in range(D)]
        self.lt = k.vd()
        self.y = [k.vd() for _ in range(D)]
        self.Lreg = {}
        for i in range(1, D):
            for kc in range(i):
                if (i, kc) not in LDS_L and (i, kc) not in GLB_L:
                    self.Lreg[(i, kc)] = k.vd()
        self.loga, self.nacc, self.sc, self.logu = k.vd(), k.vd(), k.vd(), k.vd()
        self.v_goff = [self.v_lane8]
        for j in range(1, (len(GLB_L) + 15) // 16):
            self.v_goff.append(k.v1())
            k.vop("v_add_u32_e32", self.v_goff[j], Lit(8192 * j), self.v_lane8)
        k.s_waitcnt_all()

    def next_task(self):
        """which (block, quantum) this wavefront runs next.  Plain launch (queue == NULL): block = its global number, the whole launch,
        once.  Queued launch: the grid is as large as the chip holds and its wavefronts PULL tasks from a counter (queue[0]); task n is
        quantum n / blocks of block n % blocks (quantum-major: a block's quanta are pulled a whole round of tasks apart), and a block's
        quanta chain through its state in HBM: whoever finishes quantum j of block b publishes queue[1 + b] = j + 1 behind an agent-scope
        release; whoever pulled quantum j + 1 polls that word, then acquires.  As phf_single_level.hip's queue, instruction for instruction
        what hipcc emits there (global_atomic_add sc0, global_load sc1 + s_sleep, buffer_inv sc1 / buffer_wbl2 sc1, global_store sc1)."""
        k, g = self.k, self.g
        k.comment("---- next task ----")
        self.l_task = k.new_label("task")
        self.l_end = k.new_label("end")
        l_plain, l_have = k.new_label("plain"), k.new_label("have")
        k.label(self.l_task)
        k.sop("s_mov_b64", EXEC, -1)
        w4 = k.sx(4)
        k.s_load(w4, g.kernarg, self.ARG_OFF["num_problems"])          # num_problems bpp bpp_magic total_waves
        a_np, a_bpp, a_magic, a_total = (w4.sub(i, 1) for i in range(4))
        q4 = k.sx(4)
        k.s_load(q4, g.kernarg, self.ARG_OFF["quantum"])               # quantum num_tasks blocks_magic rows_per_quantum
        a_quant, a_ntasks, a_bmagic, a_rpq = (q4.sub(i, 1) for i in range(4))
        assert self.ARG_OFF["rows_per_quantum"] == self.ARG_OFF["quantum"] + 12
        tb = k.sd()
        k.s_load(tb, g.kernarg, self.ARG_OFF["t_begin"])               # t_begin t_end
        tmp, task = k.s1(), k.s1()
        k.sop("s_cmp_eq_u64", None, self.s_queue, 0)
        k.branch("s_cbranch_scc1", l_plain)
        # -- queued: pull a task
        vt = k.v1()
        one = k.v1()
        save = k.sd()
        k.mov32(one, 1)
        k.emit("v_cmp_eq_u32_e32", [VCC], [0, self.v_lane], "valu", count="valu_int")
        k.sop("s_and_saveexec_b64", save, VCC)
        k.emit("global_atomic_add", [vt], [self.v_zero, one, self.s_queue], "vmem", suffix="sc0", mem="vm")
        k.sop("s_mov_b64", EXEC, save)
        k.readfirstlane(task, vt)
        k.sop("s_cmp_ge_u32", None, task, a_ntasks)
        k.branch("s_cbranch_scc1", self.l_end)
        self.udiv(self.s_quant, self.s_block, task, a_total, a_bmagic, tmp)      # quantum = task / blocks, block = task % blocks
        # wait for this block's previous quantum: queue[1 + block] >= quantum
        cur = k.sd()
        polls = k.s1()
        l_poll, l_ready, l_giveup = k.new_label("poll"), k.new_label("ready"), k.new_label("giveup")
        k.sop("s_lshl_b32", cur.lo(), self.s_block, 2)
        k.sop("s_mov_b32", cur.hi(), 0)
        self.add64(cur, cur, self.s_queue)
        k.sop("s_mov_b32", polls, 0)
        k.sop("s_cmp_eq_u32", None, self.s_quant, 0)
        k.branch("s_cbranch_scc1", l_ready)
        k.label(l_poll)
        k.emit("global_load_dword", [vt], [self.v_zero, cur], "vmem", suffix="offset:4 sc1", mem="vm")
        k.readfirstlane(tmp, vt)
        k.sop("s_cmp_ge_i32", None, tmp, self.s_quant)
        k.branch("s_cbranch_scc1", l_ready)
        k.raw_rec("s_sleep 16")
        k.sop("s_add_u32", polls, polls, 1)
        k.sop("s_cmp_lt_u32", None, polls, Lit(1 << 22))
        k.branch("s_cbranch_scc1", l_poll)
        # cannot happen in a correct run: give up, poison the counter so that every wavefront drains, raise the sticky fault word
        k.label(l_giveup)
        k.sop("s_lshl_b32", cur.lo(), a_total, 2)
        k.sop("s_mov_b32", cur.hi(), 0)
        self.add64(cur, cur, self.s_queue)
        k.mov32(one, 1)
        k.emit("global_store_dword", [], [self.v_zero, one, cur], "vmem", suffix="offset:4 sc1", mem="vm")
        k.mov32(one, Lit(0x40000000))
        k.emit("global_store_dword", [], [self.v_zero, one, self.s_queue], "vmem", suffix="sc1", mem="vm")
        k.branch("s_branch", self.l_end)
        k.label(l_ready)
        k.raw_rec("buffer_inv sc1")
        k.free(cur, polls, vt, one, save)
        # this quantum's iterations: (t_begin + quantum * Q, min(that + Q, t_end)]
        k.sop("s_mul_i32", tmp, self.s_quant, a_quant)
        k.sop("s_add_u32", self.s_first, tb.lo(), tmp)
        k.sop("s_add_u32", self.s_tend, self.s_first, a_quant)
        k.sop("s_min_u32", self.s_tend, self.s_tend, tb.hi())
        k.sop("s_add_u32", self.s_first, self.s_first, 1)
        k.sop("s_mul_i32", task, self.s_quant, a_rpq)             # rows saved by this block's earlier quanta (< 2^32)
        k.branch("s_branch", l_have)
        # -- plain: one task per wavefront
        k.label(l_plain)
        k.sop("s_lshl_b32", self.s_block, g.wg_id, 2)
        k.sop("s_add_u32", self.s_block, self.s_block, self.s_wave)
        k.sop("s_cmp_ge_u32", None, self.s_block, a_total)
        k.branch("s_cbranch_scc1", self.l_end)
        k.sop("s_mov_b32", self.s_quant, 0)
        k.sop("s_mov_b32", task, 0)
        k.sop("s_add_u32", self.s_first, tb.lo(), 1)
        k.sop("s_mov_b32", self.s_tend, tb.hi())
        k.label(l_have)
        k.free(q4, tb)
        k.sop("s_mov_b32", self.s_t, self.s_first)
        self.block_setup(w4, tmp, task)

    def block_setup(self, w4, tmp, task):
        """w4 = (num_problems, bpp, bpp_magic, total_waves), tmp a scalar temporary, task = rows saved by this block's earlier quanta"""
        k, g = self.k, self.g
        a_np, a_bpp, a_magic, a_total = (w4.sub(i, 1) for i in range(4))
        k.comment("---- the block: which problem, which chains; uniform area; addresses; state -> registers / LDS ----")
        a_C = k.s1()
        k.s_load(a_C, g.kernarg, self.ARG_OFF["chains"])
        # slot = block / bpp (host-checked magic), chunk = block - slot * bpp
        slot, chunk = k.s1(), k.s1()
        self.udiv(slot, chunk, self.s_block, a_bpp, a_magic, tmp)     # slot = block / bpp, chunk = block % bpp
        k.free(w4)
        # q = launch_order ? launch_order[slot] : slot
        q = k.s1()
        a_order = self.ptr("launch_order")
        k.sop("s_mov_b32", q, slot)
        l_noorder = k.new_label("noorder")
        k.sop("s_cmp_eq_u64", None, a_order, 0)
        k.branch("s_cbranch_scc1", l_noorder)
        k.sop("s_lshl_b32", tmp, slot, 2)
        k.s_load(q, a_order, tmp)
        k.label(l_noorder)
        k.free(a_order, slot)
        # pair, problem id, chain offset
        s_pair = k.s1()
        s_coff = k.s1()
        a_pi, a_pid, a_coff = self.ptr("pair_index"), self.ptr("problem_id"), self.ptr("chain_offset")
        k.sop("s_lshl_b32", tmp, q, 2)
        k.s_load(s_pair, a_pi, tmp)
        k.s_load(self.s_pid, a_pid, tmp)
        k.sop("s_mov_b32", s_coff, 0)
        l_nocoff = k.new_label("nocoff")
        k.sop("s_cmp_eq_u64", None, a_coff, 0)
        k.branch("s_cbranch_scc1", l_nocoff)
        k.s_load(s_coff, a_coff, tmp)
        k.label(l_nocoff)
        k.free(a_pi, a_pid, a_coff)
        a_stride = k.s1()
        k.s_load(a_stride, g.kernarg, self.ARG_OFF["pts_stride"])
        lane = self.v_lane
        c0 = k.s1()
        k.sop("s_lshl_b32", c0, chunk, 6)
        # the wavefront's uniform area (before EXEC shrinks to the live chains): lanes 0..11 the points, lanes 0..14 the prior
        a_lc, a_y = self.ptr("ln_conc"), self.ptr("response")
        pp = k.sd()
        k.sop("s_mul_i32", pp.lo(), s_pair, a_stride)
        k.sop("s_mul_hi_u32", pp.hi(), s_pair, a_stride)
        k.sop("s_lshl_b64", pp, pp, 3)
        self.add64(a_lc, a_lc, pp)
        self.add64(a_y, a_y, pp)
        save = k.sd()
        ua, u1, u2 = k.v1(), k.vd(), k.vd()
        k.emit("v_cmp_gt_u32_e32", [VCC], [N_PTS, lane], "valu", count="valu_int")
        k.sop("s_and_saveexec_b64", save, VCC)
        k.gload(u1, self.v_lane8, a_lc)
        k.gload(u2, self.v_lane8, a_y)
        per = SHAPE[0]
        if all(n == per for n in SHAPE) and per & (per - 1) == 0:
            k.vop("v_and_b32_e32", ua, Lit(0x100000000 - per), lane)     # point p of experiment p / per: doubles 2 per (p / per) + (p % per) [+ per for y]
            k.vop("v_add_u32_e32", ua, ua, lane)
            k.vop("v_lshl_add_u32", ua, ua, 3, self.v_ulds)
            k.ds_write(ua, u1, U_POINTS)
            k.ds_write(ua, u2, U_POINTS + 8 * per)
        else:
            # point p of experiment e (p in [PT_START[e], PT_START[e] + SHAPE[e])): ln c at double PT_OFF[e] / 8 + (p - PT_START[e]), y PT_YOFF[e] / 8
            # further on: p + a constant per experiment, built up boundary by boundary (once per task)
            uy, vtmp = k.v1(), k.v1()
            cl = [PT_OFF[e] // 8 - PT_START[e] for e in range(NE)]
            cy = [cl[e] + PT_YOFF[e] // 8 for e in range(NE)]
            k.vop("v_add_u32_e32", ua, cl[0], lane)
            k.vop("v_add_u32_e32", uy, cy[0], lane)
            for e in range(1, NE):
                if cl[e] != cl[e - 1] or cy[e] != cy[e - 1]:
                    k.cmp_u32("le", VCC, PT_START[e], lane)
                    for reg, cc in ((ua, cl), (uy, cy)):
                        if cc[e] != cc[e - 1]:
                            assert 0 < cc[e] - cc[e - 1] <= 64
                            k.mov32(vtmp, cc[e] - cc[e - 1])
                            k.cnd32_vcc(vtmp, 0, vtmp)
                            k.vop("v_add_u32_e32", reg, reg, vtmp)
            k.vop("v_lshl_add_u32", ua, ua, 3, self.v_ulds)
            k.vop("v_lshl_add_u32", uy, uy, 3, self.v_ulds)
            k.ds_write(ua, u1, U_POINTS)
            k.ds_write(uy, u2, U_POINTS)
            k.free(uy, vtmp)
        k.sop("s_mov_b64", EXEC, save)
        k.emit("v_cmp_gt_u32_e32", [VCC], [15, lane], "valu", count="valu_int")
        k.sop("s_and_saveexec_b64", save, VCC)
        k.gload(u1, self.v_lane8, g.kernarg, self.ARG_OFF["prior_loc"])
        k.vop("v_add_u32_e32", ua, self.v_ulds, self.v_lane8)
        k.ds_write(ua, u1, U_LOC)
        k.sop("s_mov_b64", EXEC, save)
        k.free(save, ua, u1, u2, a_lc, a_y, pp, s_pair, a_stride)
        more = k.sx(4)
        k.s_load(more, g.kernarg, self.ARG_OFF["seed_lo"])             # seed_lo seed_hi chain_id_base pts_stride
        assert self.ARG_OFF["prior_inv_scale"] == self.ARG_OFF["prior_loc"] + 40 and self.ARG_OFF["prior_shape_m1"] == self.ARG_OFF["prior_loc"] + 80
        # lanes: c = chunk * 64 + lane < C
        k.sop("s_sub_u32", tmp, a_C, c0)
        k.cmp_u32("gt", VCC, tmp, lane)
        k.sop("s_and_b64", EXEC, EXEC, VCC)
        k.sop("s_add_u32", tmp, more.sub(2, 1), s_coff)           # chain_id_base + chain_offset[q] + chunk * 64 + lane
        k.sop("s_add_u32", tmp, tmp, c0)
        k.vop("v_add_u32_e32", self.v_cid, tmp, lane)
        k.free(s_coff, chunk)
        # g0 = q * C + chunk * 64: first chain of this wavefront in the [.][Q * C] arrays
        k.sop("s_mul_i32", self.s_g0, q, a_C)
        k.sop("s_add_u32", self.s_g0, self.s_g0, c0)
        # nch8 = Q * C * 8 (64 bits)
        a_np = k.s1()
        k.s_load(a_np, g.kernarg, self.ARG_OFF["num_problems"])
        k.sop("s_mul_i32", self.s_nch8.lo(), a_np, a_C)
        k.sop("s_mul_hi_u32", self.s_nch8.hi(), a_np, a_C)
        k.free(a_np)
        k.sop("s_lshl_b64", self.s_nch8, self.s_nch8, 3)
        # rows: cursor = rows + ((q * 12) * C + chunk * 64) * 8 + (quantum * rows_per_quantum) * stride; stride per saved row = 12 * nch8
        # flags: bit 0 rows, bit 1 moments
        t2 = k.sd()
        a_rows, a_mom = self.ptr("rows"), self.ptr("moments")
        k.sop("s_cmp_lg_u64", None, a_rows, 0)
        k.sop("s_cselect_b32", self.s_flags, 1, 0)
        k.sop("s_cmp_lg_u64", None, a_mom, 0)
        k.sop("s_cselect_b32", tmp, 2, 0)
        k.sop("s_or_b32", self.s_flags, self.s_flags, tmp)
        k.sop("s_mul_i32", tmp, q, a_C)                           # q C  (< 2^31)
        k.sop("s_mul_hi_u32", t2.hi(), tmp, D + 1)
        k.sop("s_mul_i32", t2.lo(), tmp, D + 1)
        self.add64(t2, t2, c0)
        k.sop("s_lshl_b64", t2, t2, 3)
        self.add64(self.s_rows, a_rows, t2)
        k.sop("s_mul_hi_u32", tmp, self.s_nch8.lo(), D + 1)
        k.sop("s_mul_i32", self.s_rstride.lo(), self.s_nch8.lo(), D + 1)
        k.sop("s_mul_i32", self.s_rstride.hi(), self.s_nch8.hi(), D + 1)
        k.sop("s_add_u32", self.s_rstride.hi(), self.s_rstride.hi(), tmp)
        k.sop("s_mul_hi_u32", t2.hi(), task, self.s_rstride.lo())  # + (rows saved by this block's earlier quanta) * stride
        k.sop("s_mul_i32", t2.lo(), task, self.s_rstride.lo())
        k.sop("s_mul_i32", tmp, task, self.s_rstride.hi())
        k.sop("s_add_u32", t2.hi(), t2.hi(), tmp)
        self.add64(self.s_rows, self.s_rows, t2)
        k.sop("s_lshl_b32", self.s_c8, a_C, 3)
        # moments base = moments + g0 * 8; state; gamma
        k.sop("s_mov_b32", t2.lo(), self.s_g0)
        k.sop("s_mov_b32", t2.hi(), 0)
        k.sop("s_lshl_b64", t2, t2, 3)
        self.add64(self.s_mom, a_mom, t2)
        k.free(a_rows, a_mom)
        a_state = self.ptr("state")
        s_state = k.sd()
        self.add64(s_state, a_state, t2)
        k.free(a_state)
        k.s_load(self.s_gamma, g.kernarg, self.ARG_OFF["gamma"])
        k.free(t2, q, c0)
        # loop scalars
        until0 = k.s1()
        k.s_load(until0, g.kernarg, self.ARG_OFF["until_save0"])
        c4 = k.sx(4)
        k.s_load(c4, g.kernarg, self.ARG_OFF["adapt_start"])           # adapt_start thinning moments_after chains
        a_adapt, a_thin, a_mafter = (c4.sub(i, 1) for i in range(3))
        k.sop("s_mov_b32", self.s_adapt, a_adapt)
        k.sop("s_mov_b32", self.s_thin, a_thin)
        k.sop("s_mov_b32", self.s_mafter, a_mafter)
        k.sop("s_mov_b32", self.s_until, until0)
        k.sop("s_mov_b32", self.s_seedlo, more.sub(0, 1))
        k.sop("s_mov_b32", self.s_seedhi, more.sub(1, 1))
        k.sop("s_mov_b64", self.s_acc, 0)
        k.s_waitcnt_all()
        k.free(c4, more, until0, tmp, task, a_C)
        self.walk_state(s_state, load=True)
        k.free(s_state)

    def task_done(self):
        """state back to HBM; queued: release, publish queue[1 + block] = quantum + 1, next task"""
        k = self.k
        k.comment("---- the block's state: registers and LDS slots -> HBM; hand the block over ----")
        st, t2 = k.sd(), k.sd()
        k.s_load(st, self.g.kernarg, self.ARG_OFF["state"])
        k.sop("s_mov_b32", t2.lo(), self.s_g0)
        k.sop("s_mov_b32", t2.hi(), 0)
        k.sop("s_lshl_b64", t2, t2, 3)
        self.add64(st, st, t2)
        self.walk_state(st, load=False)
        k.free(st, t2)
        self.hand_over()

    def hand_over(self):
        """queued: release, publish queue[1 + block] = quantum + 1, next task; plain: the wavefront is done"""
        k = self.k
        t2 = k.sd()
        st = None
        fused = not getattr(self, "emit_end", True)          # the fused kernel: always queued, its task code further away than s_branch reaches
        if not fused:
            k.sop("s_cmp_eq_u64", None, self.s_queue, 0)
            k.branch("s_cbranch_scc1", self.l_end)
        k.sop("s_mov_b64", EXEC, -1)
        k.s_waitcnt_all()
        if not getattr(self, "write_through_release", False):
            k.raw_rec("buffer_wbl2 sc1")
            k.raw_rec("s_waitcnt vmcnt(0)")
        k.sop("s_lshl_b32", t2.lo(), getattr(self, "s_qblock", self.s_block), 2)
        k.sop("s_mov_b32", t2.hi(), 0)
        self.add64(t2, t2, self.s_queue)
        v = k.v1()
        tmp = k.s1()
        k.sop("s_add_u32", tmp, self.s_quant, 1)
        k.mov32(v, tmp)
        save = k.sd()
        k.emit("v_cmp_eq_u32_e32", [VCC], [0, self.v_lane], "valu", count="valu_int")
        k.sop("s_and_saveexec_b64", save, VCC)
        k.emit("global_store_dword", [], [self.v_zero, v, t2], "vmem", suffix="offset:4 sc1", mem="vm")
        k.sop("s_mov_b64", EXEC, save)
        k.free(v, tmp, save)
        if fused:
            k.long_jump(self.l_task, t2)
            k.free(t2)
            return
        k.free(t2)
        k.branch("s_branch", self.l_task)
        k.label(self.l_end)
        k.endpgm()

    def state_rows(self):
        """(state row, kind, key) in ascending row order"""
        rows = [(S_TH + i, "th", i) for i in range(D)] + [(S_LT, "lt", None)] + [(S_MEAN + i, "mean", i) for i in range(D)]
        for i in range(D):
            for kc in range(i + 1):
                rows.append((S_TRI + tri_index(i, kc), "d" if kc == i else "L", i if kc == i else (i, kc)))
        rows += [(S_LOGA, "loga", None), (S_NACC, "nacc", None)]
        assert [r for r, _, _ in rows] == list(range(2 * D + TRI + 3))
        return rows

    def walk_state(self, cur, load):
        """one pass over the rows of the state with a 64-bit cursor (row r of this wavefront: cur + r * nch8)"""
        k = self.k
        temps = []
        for r, kind, key in self.state_rows():
            reg, slot = None, None
            if kind == "th":
                reg = self.th[key]
            elif kind == "lt":
                reg = self.lt
            elif kind == "loga":
                reg = self.loga
            elif kind == "nacc":
                reg = self.nacc
            elif kind == "mean":
                slot = SLOT_MEAN + key
            elif kind == "d":
                slot = SLOT_D + key
            elif key in self.Lreg:
                reg = self.Lreg[key]
            elif key in GLB_L:
                slot = ("glb", key)
            else:
                slot = self.l_slot(*key)

            def put(s_, t_):
                if isinstance(s_, tuple):
                    self.g_store(s_[1], t_)
                else:
                    self.lds_store(s_, t_)

            if load:
                if reg is not None:
                    k.gload(reg, self.v_lane8, cur)
                else:
                    t = k.vd()
                    k.gload(t, self.v_lane8, cur)
                    temps.append((t, slot))
            else:
                scope = "sc0 sc1" if getattr(self, "write_through_release", False) else ""
                if reg is not None:
                    k.gstore(self.v_lane8, reg, cur, scope=scope)
                else:
                    t = k.vd()
                    if isinstance(slot, tuple):
                        self.g_load(t, slot[1])
                    else:
                        self.lds_load(t, slot)
                    k.gstore(self.v_lane8, t, cur, scope=scope)
                    k.free(t)
            self.add64(cur, cur, self.s_nch8)
            if load and len(temps) == 12:
                for t, s in temps:
                    put(s, t)
                    k.free(t)
                temps = []
        for t, s in temps:
            put(s, t)
            k.free(t)

    # ------------------------------------------------------------------------------------------------------------ draws
    def draws(self):
        """phf_hier_draws_k(D, cid, pid, t, seed): three Philox4x32-7 blocks (counter (cid, pid, t, b), key = seed) -> normals z[0..10]
        (word j of block b is normal 4 b + j) and log u (the last word of the last block).  The parts of the first rounds that do
        not depend on the lane run on the scalar unit; the three blocks advance round by round side by side."""
        k, c = self.k, self.c
        NB, R = (D + 3) // 4, 7
        k.comment("---- draws of iteration t: %d Philox4x32-7 blocks -> %d normals and log u ----" % (NB, D))
        s = [k.s1() for _ in range(8)]
        p1lo, p1hi, n0u, p0ulo, p0uhi, ka, kb, tx = s

        def key(r):
            """ka, kb <- round r's key words (seed + r * Weyl constants, mod 2^32)"""
            k.sop("s_add_u32", ka, self.s_seedlo, Lit((r * M.PHILOX_W0) & 0xffffffff))
            k.sop("s_add_u32", kb, self.s_seedhi, Lit((r * M.PHILOX_W1) & 0xffffffff))

        # round 0, uniform half: p1 = M1 * t; n0 = hi(p1) ^ pid ^ k0; c1' = lo(p1)
        key(0)
        k.sop("s_mul_i32", p1lo, self.s_t, c["PM1"])
        k.sop("s_mul_hi_u32", p1hi, self.s_t, c["PM1"])
        k.sop("s_xor_b32", n0u, p1hi, self.s_pid)
        k.sop("s_xor_b32", n0u, n0u, ka)
        # round 0, lane half: p0 = M0 * cid (the same for the three blocks); n2_b = hi(p0) ^ b ^ k1; c3' = lo(p0)
        P0 = k.vd()
        k.mad_u64_u32(P0, self.v_cid, c["PM0"])
        sets = [[(k.vd(), k.vd()), (k.vd(), k.vd())] for _ in range(NB)]        # per block: two (p0, p1) register pairs, used in turn
        n2 = [sets[b][0][0].hi() for b in range(NB)]
        for b in range(NB):
            k.xor3(n2[b], P0.hi(), b, kb)
        # round 1: p0' = M0 * n0 (uniform); p1'_b = M1 * n2_b; n0'_b = hi(p1'_b) ^ c1' ^ k0; n2' = hi(p0') ^ c3' ^ k1 (the same for all b)
        key(1)
        k.sop("s_mul_i32", p0ulo, n0u, c["PM0"])
        k.sop("s_mul_hi_u32", p0uhi, n0u, c["PM0"])
        k.sop("s_xor_b32", tx, p1lo, ka)
        for b in range(NB):
            k.mad_u64_u32(sets[b][1][1], n2[b], c["PM1"])
        for b in range(NB):
            k.vop("v_xor_b32_e32", sets[b][1][1].hi(), tx, sets[b][1][1].hi())       # n0'_b
        k.sop("s_xor_b32", tx, p0uhi, kb)
        n2s = P0.hi()
        k.vop("v_xor_b32_e32", n2s, tx, P0.lo())                                       # n2' (shared); c3'' = lo(p0') uniform
        # round 2: p0''_b = M0 * n0'_b; p1'' = M1 * n2' (shared); n0''_b = hi(p1'') ^ c1''_b ^ k0; n2''_b = hi(p0''_b) ^ c3'' ^ k1
        key(2)
        P1s = k.vd()
        k.mad_u64_u32(P1s, n2s, c["PM1"])
        for b in range(NB):
            k.mad_u64_u32(sets[b][0][0], sets[b][1][1].hi(), c["PM0"])
        k.sop("s_xor_b32", tx, p0ulo, kb)
        for b in range(NB):
            k.xor3(sets[b][0][1].hi(), P1s.hi(), sets[b][1][1].lo(), ka)               # n0''_b  (c1''_b = lo(p1'_b))
        for b in range(NB):
            k.vop("v_xor_b32_e32", sets[b][0][0].hi(), tx, sets[b][0][0].hi())       # n2''_b
        # from here on every word is per lane: block b's state = (n0, c1, n2, c3)
        st = [[sets[b][0][1].hi(), P1s.lo(), sets[b][0][0].hi(), sets[b][0][0].lo()] for b in range(NB)]
        for r in range(3, R):
            key(r)
            cur = r % 2
            for b in range(NB):
                k.mad_u64_u32(sets[b][cur][0], st[b][0], c["PM0"])
                k.mad_u64_u32(sets[b][cur][1], st[b][2], c["PM1"])
            for b in range(NB):
                p0, p1 = sets[b][cur]
                k.xor3(p1.hi(), p1.hi(), st[b][1], ka)
                k.xor3(p0.hi(), p0.hi(), st[b][3], kb)
                st[b] = [p1.hi(), p1.lo(), p0.hi(), p0.lo()]
        k.free(s)
        # normals: word j of block b -> z[4 b + j]; the accept uniform: word 3 of block 2
        self.zn = [k.vd() for _ in range(D)]
        for b in range(NB):
            idx = [4 * b + j for j in range(4) if 4 * b + j < D]
            for grp in (idx[:2], idx[2:]):                                         # two at a time: 15 registers each while it runs
                M.normal_u32(self.m, [self.zn[i] for i in grp], [st[b][i - 4 * b] for i in grp])
        M.unit_open32(self.m, self.logu, st[NB - 1][3])
        M.log_pos(self.m, [self.logu], [self.logu])
        k.free(P0, P1s, [list(p) for blk in sets for p in blk])

    # ------------------------------------------------------------------------------------------------------------ sweep
    def load_gamma(self):
        """v_gs <- (t - 1 > adapt_start and t > first) ? gamma[t - 1 - adapt_start] : 0, issued at the top of the loop: a vector-memory
        load (its own counter: nothing else waits for it) that the draws cover"""
        k = self.k
        self.v_gs = k.vd()
        tmp = k.s1()
        cur = k.sd()
        l_gdone = k.new_label("gdone")
        k.mov32(self.v_gs.lo(), 0)
        k.mov32(self.v_gs.hi(), 0)
        k.sop("s_cmp_eq_u32", None, self.s_t, self.s_first)
        k.branch("s_cbranch_scc1", l_gdone)
        k.sop("s_sub_u32", tmp, self.s_t, 1)
        k.sop("s_cmp_le_u32", None, tmp, self.s_adapt)
        k.branch("s_cbranch_scc1", l_gdone)
        k.sop("s_sub_u32", tmp, tmp, self.s_adapt)
        k.sop("s_lshl_b32", cur.lo(), tmp, 3)
        k.sop("s_lshr_b32", cur.hi(), tmp, 29)
        self.add64(cur, cur, self.s_gamma)
        k.gload(self.v_gs, self.v_zero, cur)
        k.label(l_gdone, drain=False)
        k.free(tmp, cur)

    def g_fetch_column(self, kc):
        """issue the loads of column kc's scratch-tier elements (self.gq: element -> register pair)"""
        for i in range(kc + 1, D):
            if (i, kc) in GLB_L:
                self.gq[(i, kc)] = self.k.vd()
                self.g_load(self.gq[(i, kc)], (i, kc))

    def sweep_prefetch(self):
        """top of the loop, ahead of the draws: the scratch-tier elements of the sweep's first columns (a device-memory latency that the
        draws cover)"""
        self.gq = {}
        if G_STREAM:
            self.g_fifo, self.g_next = [], 0
            self.g_top_up()
            return
        self.g_fetched = min(G_AHEAD, G_PRE) if GLB_L else 0        # (the draws leave room for G_PRE columns; the rest of the lead is taken at the sweep's start)
        for kc in range(self.g_fetched):
            self.g_fetch_column(kc)

    def g_top_up(self):
        """the streamed scratch tier: keep G_STREAM loads in flight over GLB_L's order"""
        while len(self.g_fifo) < G_STREAM and self.g_next < len(GLB_L):
            r = self.k.vd()
            self.g_load(r, GLB_L[self.g_next])
            self.g_fifo.append((GLB_L[self.g_next], r))
            self.g_next += 1

    def sweep_column_streamed(self, kc, w, y, lq, hd, head, inv, beta, sq, pos, alpha, omg):
        """column kc of the sweep with the scratch tier streamed (G_STREAM): the same operations per value as the column-at-once form — w_i <- w_i -
        w_k L_ik with the OLD L_ik, L_ik <- L_ik + beta w_i with the new w_i, y_i += L_ik u_k column after column — only their order ACROSS elements
        differs: an element of the scratch tier gets all three once beta is known, four elements side by side"""
        k, m = self.k, self.m
        dk, ap, dn = hd["dk"], hd["ap"], hd["dn"]
        near = [i for i in range(kc + 1, D) if (i, kc) not in GLB_L]          # registers / LDS: as in the other form
        far = [i for i in range(kc + 1, D) if (i, kc) in GLB_L]
        lik = {i: (lq[(i, kc)] if (i, kc) in LDS_L else self.Lreg[(i, kc)]) for i in near}
        with k.parallel() as par:
            par.stream()
            M.rcp(m, [inv], [dn])
            par.stream()
            M.sqrt_nonneg(m, [sq], [dn], [pos])
            k.mul(self.zn[kc], sq, self.zn[kc])
            if near:
                par.stream()
                for i in near:
                    k.fma(w[i], Neg(w[kc]), lik[i], w[i])
        k.cnd64(inv, 0.0, inv, pos)
        k.mul(beta, ap, inv)
        k.mul(dk, alpha, dk)
        k.mul(dk, dk, inv)
        k.cnd64(alpha, alpha, dk, pos)
        k.free(dk, ap, dn, inv, sq)
        for i in near:
            k.fma(lik[i], beta, w[i], lik[i])
            if (i, kc) in LDS_L:
                self.lds_store(self.l_slot(i, kc), lik[i])
            k.fma(y[i], lik[i], self.zn[kc], 0.0 if kc == 0 else y[i])
            if (i, kc) in LDS_L:
                k.free(lq.pop((i, kc)))
        k.add(y[kc], 0.0 if kc == 0 else y[kc], self.zn[kc])
        for g0 in range(0, len(far), 4):
            grp = far[g0:g0 + 4]
            regs = []
            for i in grp:
                key, r = self.g_fifo.pop(0)
                assert key == (i, kc), (key, i, kc)
                regs.append(r)
            for i, r in zip(grp, regs):
                k.fma(w[i], Neg(w[kc]), r, w[i])
            for i, r in zip(grp, regs):
                k.fma(r, beta, w[i], r)
            for i, r in zip(grp, regs):
                self.g_store((i, kc), r)
                k.fma(y[i], r, self.zn[kc], 0.0 if kc == 0 else y[i])
            k.free(regs)
            self.g_top_up()
        k.free(beta)
        nxt = head(kc + 1) if kc + 1 < D else None
        k.free(w[kc], self.zn[kc])
        return nxt

    def sweep(self):
        """the adaptation of iteration t - 1 (hier_advance_body: mean, loga, PHF_LDL_COLUMN column by column) fused with
        y = L sqrt(d) z of iteration t's proposal; v_gs = gamma (0.0 where the C code does not adapt: an exact no-op).
        Column k runs as: [1 / dn chain | sqrt(dn) chain | w_i -= w_k L_ik] side by side, then [L_ik, y_i updates | head of column k + 1]."""
        k, c, m = self.k, self.c, self.m
        k.comment("---- adaptation of t - 1 (rank-one update of L D L') + y = L sqrt(d) z for t ----")
        gs = self.v_gs
        omg, alpha = k.vd(), k.vd()
        k.add(omg, 1.0, Neg(gs))
        # w = theta - mean; mean <- g theta + (1 - g) mean
        w = [k.vd() for _ in range(D)]
        dq = {}
        lq = {}

        def fetch_column(kc):
            for i in range(kc + 1, D):
                if (i, kc) in LDS_L:
                    lq[(i, kc)] = k.vd()
                    self.lds_load(lq[(i, kc)], self.l_slot(i, kc))

        def mean_update(rows, mt):
            for j, i in enumerate(rows):
                k.sub(w[i], self.th[i], mt[j])
                k.mul(mt[j], omg, mt[j])
                k.fma(mt[j], gs, self.th[i], mt[j])
                self.lds_store(SLOT_MEAN + i, mt[j])

        if not GLB_L:
            mt = [k.vd() for _ in range(D)]
            for i in range(D):
                self.lds_load(mt[i], SLOT_MEAN + i)
            for kc in range(2):                              # the first diagonals: in flight while the mean is updated
                dq[kc] = k.vd()
                self.lds_load(dq[kc], SLOT_D + kc)
            fetch_column(0)
            mean_update(range(D), mt)
            k.free(mt)
        else:
            # (register budget: the mean goes through five doubles at a time, the next group's reads in flight behind the current one's arithmetic)
            groups = [list(range(i0, min(i0 + 5, D))) for i0 in range(0, D, 5)]
            mts = []
            for gi, rows in enumerate(groups[:2]):
                mts.append([k.vd() for _ in rows])
                for j, i in enumerate(rows):
                    self.lds_load(mts[gi][j], SLOT_MEAN + i)
            for gi, rows in enumerate(groups):
                mean_update(rows, mts[gi])
                k.free(mts[gi])
                if gi + 2 < len(groups):
                    nxt = groups[gi + 2]
                    mts.append([k.vd() for _ in nxt])
                    for j, i in enumerate(nxt):
                        self.lds_load(mts[gi + 2][j], SLOT_MEAN + i)
                if gi == len(groups) - 2:
                    for kc in range(2):
                        dq[kc] = k.vd()
                        self.lds_load(dq[kc], SLOT_D + kc)
                    fetch_column(0)
        # loga <- loga + g ((accepted ? 1 : 0) - 1/4)
        a01 = k.vd()
        k.mov32(a01.lo(), 0)
        k.cnd32(a01.hi(), 0, c["ONEHI"], self.s_acc)
        k.sub(a01, a01, c["QUARTER"])
        k.fma(self.loga, gs, a01, self.loga)
        k.free(a01)
        k.mov64(alpha, gs)
        k.free(gs)
        y = self.y
        pos = k.sd()

        def head(kc):
            """dk = (1 - g) d_k; ap = alpha w_k; dn = fma(ap, w_k, dk): what column kc's chains start from"""
            h = {"dk": k.vd(), "ap": k.vd(), "dn": k.vd()}
            k.mul(h["dk"], omg, dq[kc])
            k.mul(h["ap"], alpha, w[kc])
            k.fma(h["dn"], h["ap"], w[kc], h["dk"])
            self.lds_store(SLOT_D + kc, h["dn"])
            k.free(dq.pop(kc))
            return h

        def source(i, kc):
            if (i, kc) in LDS_L:
                return lq[(i, kc)]
            if (i, kc) in GLB_L:
                return self.gq[(i, kc)]
            return self.Lreg[(i, kc)]

        hd = head(0)
        for kc in range(D):
            if kc + 2 < D:
                dq[kc + 2] = k.vd()
                self.lds_load(dq[kc + 2], SLOT_D + kc + 2)
            if kc + 1 < D:
                fetch_column(kc + 1)
            while GLB_L and not G_STREAM and self.g_fetched <= kc + G_AHEAD and self.g_fetched < D:
                self.g_fetch_column(self.g_fetched)
                self.g_fetched += 1
            dk, ap, dn = hd["dk"], hd["ap"], hd["dn"]
            inv, beta, sq = k.vd(), k.vd(), k.vd()
            if G_STREAM:
                hd = self.sweep_column_streamed(kc, w, y, lq, hd, head, inv, beta, sq, pos, alpha, omg)
                continue
            lik = {i: source(i, kc) for i in range(kc + 1, D)}
            with k.parallel() as par:
                par.stream()                                 # 1 / dn (or 0), beta, the next alpha
                M.rcp(m, [inv], [dn])
                par.stream()                                 # sqrt(dn) (or 0), u_k = sqrt(dn) z_k
                M.sqrt_nonneg(m, [sq], [dn], [pos])
                k.mul(self.zn[kc], sq, self.zn[kc])
                par.stream()                                 # w_i <- w_i - w_k L_ik: needs w_k only
                for i in range(kc + 1, D):
                    k.fma(w[i], Neg(w[kc]), lik[i], w[i])
            k.cnd64(inv, 0.0, inv, pos)
            k.mul(beta, ap, inv)
            k.mul(dk, alpha, dk)
            k.mul(dk, dk, inv)
            k.cnd64(alpha, alpha, dk, pos)
            k.free(dk, ap, dn, inv, sq)
            with k.parallel() as par:
                par.stream()                                 # L_ik <- L_ik + beta w_i; y_i += L_ik u_k
                for i in range(kc + 1, D):
                    k.fma(lik[i], beta, w[i], lik[i])
                for i in range(kc + 1, D):
                    if (i, kc) in LDS_L:
                        self.lds_store(self.l_slot(i, kc), lik[i])
                    elif (i, kc) in GLB_L:
                        self.g_store((i, kc), lik[i])
                    k.fma(y[i], lik[i], self.zn[kc], 0.0 if kc == 0 else y[i])
                    if (i, kc) in LDS_L:
                        k.free(lq.pop((i, kc)))
                    elif (i, kc) in GLB_L:
                        k.free(self.gq.pop((i, kc)))
                k.add(y[kc], 0.0 if kc == 0 else y[kc], self.zn[kc])
                if kc + 1 < D:
                    par.stream()                             # the head of the next column (alpha and w_{k+1} are final)
                    hd = head(kc + 1)
            k.free(beta)
            if GLB_L:                                        # column kc is done with w_k and u_k: their registers take later columns' prefetches
                k.free(w[kc], self.zn[kc])
        if GLB_L:
            k.free(pos, omg, alpha)
        else:
            k.free(pos, omg, alpha, w, self.zn)
        # the scale of the next proposal: e^(loga / 2)
        h = k.vd()
        k.mul(h, self.loga, 0.5)
        M.exp_fast(m, [self.sc], [h])
        k.free(h)

    # ------------------------------------------------------------------------------------------------------------ save
    def save(self):
        """thinning + sample store + moments of iteration t - 1 (hier_advance_body, :502-503 of the reference)"""
        k = self.k
        k.comment("---- save iteration t - 1: thinned sample, moments ----")
        l_skip = k.new_label("nosave")
        k.sop("s_cmp_eq_u32", None, self.s_t, self.s_first)
        k.branch("s_cbranch_scc1", l_skip)
        k.sop("s_sub_u32", self.s_until, self.s_until, 1)
        k.sop("s_cmp_lg_u32", None, self.s_until, 0)
        k.branch("s_cbranch_scc1", l_skip)
        k.sop("s_mov_b32", self.s_until, self.s_thin)
        l_norows = k.new_label("norows")
        k.sop("s_bitcmp0_b32", None, self.s_flags, 0)
        k.branch("s_cbranch_scc1", l_norows)
        cur = k.sd()
        k.sop("s_mov_b64", cur, self.s_rows)
        for i in range(D + 1):
            k.gstore(self.v_lane8, self.th[i] if i < D else self.lt, cur)
            if i < D:
                self.add64(cur, cur, self.s_c8)
        self.add64(self.s_rows, self.s_rows, self.s_rstride)
        k.label(l_norows)
        k.sop("s_bitcmp0_b32", None, self.s_flags, 1)
        k.branch("s_cbranch_scc1", l_skip)
        tmp = k.s1()
        k.sop("s_sub_u32", tmp, self.s_t, 1)
        k.sop("s_cmp_le_u32", None, tmp, self.s_mafter)
        k.branch("s_cbranch_scc1", l_skip)
        k.free(tmp)
        k.sop("s_mov_b64", cur, self.s_mom)
        sq = [k.vd() for _ in range(4)]
        for r in range(2 * (D + 1)):
            i = r % (D + 1)
            src = self.th[i] if i < D else self.lt
            if r >= D + 1:
                t = sq[r % 4]
                k.mul(t, src, src)
                src = t
            k.gatomic_add_f64(self.v_lane8, src, cur, scope="sc1" if getattr(self, "write_through_release", False) else "")
            self.add64(cur, cur, self.s_nch8)
        k.free(sq, cur)
        k.label(l_skip)

    # ------------------------------------------------------------------------------------------------------------ target
    def target(self, lt_star):
        """phf_hier_log_target_n(3, PHF_HIER_SHAPE(4, 0), ...) of the proposal in self.y (= star): both halves, experiment by experiment"""
        k, c, m = self.k, self.c, self.m
        st = self.y
        alpha, beta, mu, s_, sigma = st[0], st[1], st[2], st[3], st[D - 1]
        pic = [st[4 + 2 * i] for i in range(NE)]
        hill = [st[5 + 2 * i] for i in range(NE)]
        k.comment("---- target of the proposal: support, the 12 + 4 logarithms, the 12 points ----")
        # support (phf_hier_out_of_support) and xl = hv - loc
        s_bad = k.sd()
        # the prior's 15 doubles (loc, 1 / scale, shape - 1) -> registers, all in flight before the first use: seven 16-byte reads, one 8-byte
        pq = [self.uload(k.vq(), U_LOC + 16 * n) for n in range(7)] + [self.uload(k.vd(), U_LOC + 112)]
        pd = [pq[n // 2].sub(2 * (n % 2)) if n < 14 else pq[7] for n in range(15)]
        loc, isc, sm1 = pd[0:5], pd[5:10], pd[10:15]
        f_loc, f_isc, f_sm1 = pq[0:2], pq[2:5], pq[5:8]
        hv = [alpha, beta, mu, s_, sigma]
        for j in range(5):
            k.cmp("le", VCC, hv[j], loc[j])
            k.sop("s_mov_b64" if j == 0 else "s_or_b64", s_bad, *( [VCC] if j == 0 else [s_bad, VCC]))
        for i in range(NE):
            k.cmp("lt", VCC, hill[i], 0.0)
            k.sop("s_or_b64", s_bad, s_bad, VCC)
            k.cmp("lt", VCC, pic[i], -2.0)
            k.sop("s_or_b64", s_bad, s_bad, VCC)
        xl = [k.vd() for _ in range(5)]
        for j in range(5):
            k.sub(xl[j], hv[j], loc[j])
        # 1 / sigma and 1 / s behind one reciprocal (phf_batch_recip of {sigma, s})
        inv_s, inv_sc, pre, inv = k.vd(), k.vd(), k.vd(), k.vd()
        k.mul(pre, sigma, s_)
        M.rcp(m, [inv], [pre])
        k.mul(inv_sc, inv, sigma)
        k.mul(inv_s, inv, s_)
        k.free(pre, inv)
        # lin0 = sum_k -xl_k / scale_k  (half 0);  z_i, lin1 (half 1)
        k.free(f_loc)
        lin0, lin1 = k.vd(), k.vd()
        for j in range(5):
            k.fma(lin0, Neg(xl[j]), isc[j], 0.0 if j == 0 else lin0)
        z = [k.vd() for _ in range(NE)]
        for i in range(NE):
            k.sub(z[i], pic[i], mu)
            k.mul(z[i], z[i], inv_sc)
        for i in range(NE):
            k.sub(lin1, 0.0 if i == 0 else lin1, z[i])
        k.free(inv_sc)
        # weights that are computed: -Ne beta, beta - 1
        k.free(f_isc)
        three, twelve = k.sd(), k.sd()                       # (double)Ne and (double)n_pts: literals (their low words are zero)
        for r, val in ((three, NE), (twelve, N_PTS)):
            lo, hi = dbl_words(val)
            assert lo == 0
            k.sop("s_mov_b32", r.lo(), 0)
            k.sop("s_mov_b32", r.hi(), Lit(hi))
        wnb, wb1 = k.vd(), k.vd()
        k.mul(wnb, Neg(three), beta)
        k.add(wb1, beta, -1.0)
        # half 0: ln alpha, ln Hill_1..3, ln beta, ln(alpha - loc0)
        part0, part1 = k.vd(), k.vd()
        # (the 9 + Ne logarithms in phf_hier_target_half's order — alpha | Hill_i | beta | x_k - loc_k | s | sigma [| pad] — half 0 takes the first
        # (10 + Ne) / 2: up to alpha - loc0 for Ne = 3, 4, up to beta for Ne = 5, 6)
        assert NE in (3, 4, 5, 6)
        lg0 = [k.vd() for _ in range(NE + 1)]
        args0 = [alpha] + hill
        for j in range(0, NE + 1, 2):
            M.log_fast(m, lg0[j:j + 2], args0[j:j + 2])
        k.fma(part0, wnb, lg0[0], 0.0)
        for i in range(NE):
            k.fma(part0, wb1, lg0[1 + i], part0)
        k.free(wnb, wb1)
        t2 = [k.vd(), k.vd()]
        if NE <= 4:
            M.log_fast(m, t2, [beta, xl[0]])
            k.fma(part0, three, t2[0], part0)
            k.fma(part0, sm1[0], t2[1], part0)
            # half 1: ln(beta - loc1), ln(mu - loc2), ln(s - loc3), ln(sigma - loc4), ln s, ln sigma
            t4 = [k.vd() for _ in range(2)]
            M.log_fast(m, t2, [xl[1], xl[2]])
            M.log_fast(m, t4, [xl[3], xl[4]])
            for j, r in enumerate(t2 + t4):
                k.fma(part1, sm1[1 + j], r, 0.0 if j == 0 else part1)
            M.log_fast(m, t2, [s_, sigma])
            k.fma(part1, Neg(three), t2[0], part1)
            k.fma(part1, Neg(twelve), t2[1], part1)
        else:
            M.log_fast(m, t2[:1], [beta])
            k.fma(part0, three, t2[0], part0)
            # half 1: ln(x_k - loc_k) for all five, ln s, ln sigma
            t4 = []
            args1 = list(xl) + [s_, sigma]
            wts1 = list(sm1) + [Neg(three), Neg(twelve)]
            for j in range(0, 7, 2):
                M.log_fast(m, t2[:len(args1[j:j + 2])], args1[j:j + 2])
                for jj in range(len(args1[j:j + 2])):
                    k.fma(part1, wts1[j + jj], t2[jj], 0.0 if j + jj == 0 else part1)
        if (9 + NE) % 2:                                     # the pad of half 1: fma(0, ln 1, part) with ln 1 = +0 exactly
            k.add(part1, part1, 0.0)
        k.free(t4, xl, f_sm1, three, twelve)
        # la0_i = 1 + (Hill_i / alpha)^beta = 1 + exp(beta (ln Hill_i - ln alpha));  la1_i = 1 + exp(-z_i)
        la0, la1 = [k.vd() for _ in range(NE)], [k.vd() for _ in range(NE)]
        ea = [k.vd() for _ in range(NE)]
        for i in range(NE):
            k.sub(ea[i], lg0[1 + i], lg0[0])
            k.mul(ea[i], beta, ea[i])
        k.free(lg0)
        for i in range(NE):
            k.fmax(z[i], Neg(z[i]), c["M746"])                   # phf_exp_fast_k(-z): the clamp's first half takes the negation
            k.fmin(z[i], z[i], c["P710"])
        M.exp_fast(m, la0, ea)
        M.exp_core(m, la1, z)
        k.free(ea, z)
        for i in range(NE):
            k.add(la0[i], la0[i], 1.0)
            k.add(la1[i], la1[i], 1.0)
        k.add(part0, part0, lin0)
        k.add(part1, part1, lin1)
        k.free(lin0, lin1)
        # -2 ln prod_i la_i, term by term where the product is not below 2^1000 (rare: a uniform branch)
        prod = t2
        for h, la in ((0, la0), (1, la1)):
            k.mul(prod[h], la[0], la[1])
            for i in range(2, NE):
                k.mul(prod[h], prod[h], la[i])
        vlog = [k.vd(), k.vd()]
        big = [k.sd(), k.sd()]
        for h in range(2):
            k.cmp_lit("ngt", 0x7e700000, prod[h])                # !(prod < 2^1000)  ==  !(2^1000 > prod)
            k.sop("s_mov_b64", big[h], VCC)
        M.log_pos(m, vlog, prod)
        anyb = k.sd()
        l_nobig = k.new_label("nobig")
        k.sop("s_or_b64", anyb, big[0], big[1])
        k.sop("s_cmp_eq_u64", None, anyb, 0)
        k.branch("s_cbranch_scc1", l_nobig)
        for h, la in ((0, la0), (1, la1)):
            vs, vi = k.vd(), k.vd()
            for i in range(NE):
                M.log_pos(m, [vi], [la[i]])
                k.cmp_lit("lt", 0x7e700000, la[i])                 # la > 2^1000
                k.cnd32_vcc(vi.lo(), vi.lo(), self.v_zero)
                k.cnd32_vcc(vi.hi(), vi.hi(), self.v_infhi)
                k.add(vs, 0.0 if i == 0 else vs, vi)
            k.cnd64(vlog[h], vlog[h], vs, big[h])
            k.free(vs, vi)
        k.label(l_nobig)
        k.free(anyb, big, la0, la1)
        k.fma(part0, -2.0, vlog[0], part0)
        k.fma(part1, -2.0, vlog[1], part1)
        k.free(vlog, prod)
        # the points (:117-125): experiment i's n points, the first 2 floor((n + 2) / 4) for half 0, the others for half 1; a half takes its
        # points two at a time behind one reciprocal, then the odd one (phf_hier_target_half) — here a pair of half 0 and a pair of half 1
        # advance side by side where both exist (four points per experiment: exactly that)
        sse, mass = [k.vd(), k.vd()], [k.vd(), k.vd()]
        started = {"sse": [False, False], "mass": [False, False]}

        def acc_sse(h, r):
            k.fma(sse[h], r, r, sse[h] if started["sse"][h] else 0.0)
            started["sse"][h] = True

        def acc_mass(h, v):
            k.mul(mass[h], mass[h] if started["mass"][h] else 1.0, v)
            started["mass"][h] = True

        for i in range(NE):
            n = SHAPE[i]
            nf = min(2 * ((n + 2) // 4), n)
            units = []                                       # (half, points): a pair or a single, in each half's own order
            for h, idx in ((0, list(range(nf))), (1, list(range(nf, n)))):
                units += [(h, idx[j:j + 2]) for j in range(0, len(idx) - 1, 2)]
                if len(idx) % 2:
                    units.append((h, idx[-1:]))
            pairs0 = [u for u in units if u[0] == 0 and len(u[1]) == 2]
            pairs1 = [u for u in units if u[0] == 1 and len(u[1]) == 2]
            chunks = []
            while pairs0 and pairs1:                         # a pair of each half side by side
                chunks.append([pairs0.pop(0), pairs1.pop(0)])
            chunks += [[u] for u in pairs0 + pairs1]
            singles = [u for u in units if len(u[1]) == 1]
            chunks += [[u] for u in singles]
            # (a half's pairs come before its single, and chunks keep each half's order: its sums see the C order)
            lnic = None
            for ci, chunk in enumerate(chunks):
                pts = [p_ for _, ps in chunk for p_ in ps]
                # this chunk's points: ln c, then y (used some fifty instructions further down); a pair is one 16-byte read
                lreg, yreg, held = {}, {}, {0: [], 1: []}
                for which, (base_off, regs) in enumerate(((0, lreg), (PT_YOFF[i], yreg))):
                    for _, ps in chunk:
                        r = k.vq() if len(ps) == 2 else k.vd()
                        assert len(ps) == 1 or ps[0] % 2 == 0
                        self.uload(r, U_POINTS + PT_OFF[i] + base_off + 8 * ps[0])
                        held[which].append(r)
                        for j, p_ in enumerate(ps):
                            regs[p_] = r.sub(2 * j) if len(ps) == 2 else r
                if lnic is None:
                    lnic = k.vd()
                    k.sub(lnic, c["P6"], pic[i])
                    k.mul(lnic, c["LN10"], lnic)
                x = [k.vd() for _ in pts]
                d = [k.vd() for _ in pts]
                for j, p_ in enumerate(pts):
                    k.sub(x[j], lreg[p_], lnic)
                    k.mul(x[j], hill[i], x[j])
                    k.fmin(x[j], x[j], c["P40"])
                if ci == len(chunks) - 1:
                    k.free(lnic)
                k.free(held[0])
                M.exp_capped(m, d, x)
                for j in range(len(pts)):
                    k.add(d[j], d[j], 1.0)
                nu = len(chunk)
                inv = x[:nu]
                pred = x[nu:] + [k.vd() for _ in range(nu)]
                den, pos = [], 0
                for u, (h, ps) in enumerate(chunk):
                    if len(ps) == 2:
                        k.mul(pred[u], d[pos], d[pos + 1])
                        den.append(pred[u])
                    else:
                        den.append(d[pos])
                    pos += len(ps)
                M.rcp(m, inv, den)
                # pair: pred0 = fma(-100, inv * d1, 100), pred1 = fma(-100, inv * d0, 100); single: pred = fma(-100, inv, 100)
                tt, pos = [], 0
                for u, (h, ps) in enumerate(chunk):
                    if len(ps) == 2:
                        t0, t1 = k.vd(), k.vd()
                        k.mul(t0, inv[u], d[pos + 1])
                        k.mul(t1, inv[u], d[pos])
                        tt += [t0, t1]
                    else:
                        tt.append(inv[u])
                    pos += len(ps)
                k.free(d)
                for j in range(len(pts)):
                    k.fma(pred[j], Neg(c["K100"]), tt[j], c["K100"])
                for j, p_ in enumerate(pts):
                    if tt[j] in inv:                             # a single's w sits in inv: its residual needs a register of its own
                        tt[j] = k.vd()
                    k.sub(tt[j], yreg[p_], pred[j])
                pos = 0
                for h, ps in chunk:
                    for j in range(len(ps)):
                        acc_sse(h, tt[pos + j])
                    pos += len(ps)
                k.free([t for t in tt if t not in inv], inv, held[1])
                # truncation masses (phf_trunc_mass_x2: upper tails skipped unless some lane needs one; phf_trunc_mass for the odd point)
                pos = 0
                for h, ps in chunk:
                    pp = pred[pos:pos + len(ps)]
                    pos += len(ps)
                    ya = [k.vd() for _ in pp]
                    yb = [k.vd() for _ in pp]
                    for j, p_ in enumerate(pp):
                        k.mul(ya[j], p_, inv_s)
                        k.mul(ya[j], ya[j], c["ISQRT2"])
                        k.sub(yb[j], c["K100"], p_)
                        k.mul(yb[j], yb[j], inv_s)
                        k.mul(yb[j], yb[j], c["ISQRT2"])
                    tl = list(pp)                                        # the predictions are spent: their registers take the tails
                    M.erfc_tab(m, tl, ya)
                    k.free(ya)
                    if len(pp) == 2:
                        need = k.sd()
                        l_skip = k.new_label("noupper")
                        k.cmp("lt", need, yb[0], c["P6"])
                        k.cmp("lt", VCC, yb[1], c["P6"])
                        k.sop("s_or_b64", need, need, VCC)
                        k.sop("s_cmp_eq_u64", None, need, 0)
                        k.branch("s_cbranch_scc1", l_skip)
                        tu = [k.vd(), k.vd()]
                        M.erfc_tab(m, tu, yb)
                        k.add(tl[0], tl[0], tu[0])
                        k.add(tl[1], tl[1], tu[1])
                        k.free(tu)
                        k.label(l_skip)
                        k.free(need, yb)
                        k.fma(tl[0], -0.5, tl[0], 1.0)
                        k.fma(tl[1], -0.5, tl[1], 1.0)
                        k.mul(tl[0], tl[0], tl[1])
                    else:
                        tu = [k.vd()]
                        M.erfc_tab(m, tu, yb)
                        k.add(tl[0], tl[0], tu[0])
                        k.free(tu, yb)
                        k.fma(tl[0], -0.5, tl[0], 1.0)
                    acc_mass(h, tl[0])
                k.free(pred)
        for h in range(2):                                   # a half without points: SSE 0, mass 1
            if not started["sse"][h]:
                k.mov64(sse[h], 0.0)
            if not started["mass"][h]:
                k.mov64(mass[h], 1.0)
        # halves: part_h - fma(sse_h, (0.5 inv_s) inv_s, ln mass_h), -inf if the mass underflowed; sum; -inf outside the support
        hs = k.vd()
        k.mul(hs, inv_s, 0.5)
        k.mul(hs, hs, inv_s)
        tr = [k.vd(), k.vd()]
        M.log_pos(m, tr, mass)
        for h, part in ((0, part0), (1, part1)):
            k.fma(tr[h], sse[h], hs, tr[h])
            k.sub(part, part, tr[h])
            k.cmp_lit("ngt", 0x00100000, mass[h])               # !(mass < 2^-1022)
            k.cnd32_vcc(part.lo(), 0, part.lo())
            k.cnd32_vcc(part.hi(), c["NINFHI"], part.hi())
        k.add(lt_star, part0, part1)
        k.sop("s_not_b64", VCC, s_bad)
        k.cnd32_vcc(lt_star.lo(), 0, lt_star.lo())
        k.cnd32_vcc(lt_star.hi(), c["NINFHI"], lt_star.hi())
        k.free(hs, tr, sse, mass, part0, part1, inv_s, s_bad)

    # ------------------------------------------------------------------------------------------------------------ the kernel
    def build(self):
        k, c = self.k, self.c
        self.prologue_once()
        self.next_task()
        self.emit_loop()
        lines_meta = k.finish(LDS_BYTES, ARG_BYTES)
        return lines_meta, self.collect_info()

    def emit_loop(self):
        """the block's iterations (draws -> sweep -> save -> [leave] -> propose -> target -> accept), then the state back and the hand-over"""
        k, c = self.k, self.c
        k.count_marker("prologue")
        l_loop, l_exit = k.new_label("loop"), k.new_label("exit")
        k.label(l_loop)
        self.load_gamma()
        self.sweep_prefetch()
        self.draws()
        k.count_marker("draws")
        self.sweep()
        k.count_marker("sweep")
        self.save()
        k.count_marker("save")
        k.sop("s_cmp_gt_u32", None, self.s_t, self.s_tend)
        k.branch("s_cbranch_scc1", l_exit)
        k.comment("---- proposal theta* = theta + e^(loga/2) y (in y's registers), target, accept ----")
        for i in range(D):
            k.fma(self.y[i], self.sc, self.y[i], self.th[i])
        lt_star = k.vd()
        self.target(lt_star)
        k.count_marker("target")
        diff = k.vd()
        k.sub(diff, lt_star, self.lt)
        k.cmp("lt", self.s_acc, self.logu, diff)
        k.free(diff)
        for i in range(D):
            k.cnd64(self.th[i], self.th[i], self.y[i], self.s_acc)
        k.cnd64(self.lt, self.lt, lt_star, self.s_acc)
        k.free(lt_star)
        one = k.vd()
        k.mov32(one.lo(), 0)
        k.cnd32(one.hi(), 0, c["ONEHI"], self.s_acc)
        k.add(self.nacc, self.nacc, one)
        k.free(one)
        k.sop("s_add_u32", self.s_t, self.s_t, 1)
        k.count_marker("accept")
        k.branch("s_branch", l_loop)
        k.label(l_exit)
        self.task_done()

    def collect_info(self):
        k = self.k
        snaps = k.snapshots
        order = ["prologue", "draws", "sweep", "save", "target", "accept"]
        prev = snaps["prologue"]
        for name in order[1:]:
            cur = snaps[name]
            self.info["count_" + name] = {key: cur.get(key, 0) - prev.get(key, 0) for key in sorted(cur) if cur.get(key, 0) - prev.get(key, 0)}
            prev = cur
        tot = {}
        for name in order[1:]:
            for key, v in self.info["count_" + name].items():
                tot[key] = tot.get(key, 0) + v
        self.info["count_iteration"] = tot
        self.info["vgpr_high_water"] = k.v.high
        self.info["sgpr_high_water"] = k.s.high
        self.info["lds_bytes_per_workgroup"] = LDS_BYTES
        self.info["lds_slots_per_wavefront"] = NSLOTS
        self.info["scratch_slots_per_wavefront"] = len(GLB_L)
        return self.info


# ---------------------------------------------------------------------------------------------------------------------------------------
# ONE kernel with a body per (experiments, shape): a persistent grid pulls the blocks of EVERY launch group of a hierarchical run from one
# work queue.  Separate kernels cannot share a chip well — each persistent grid takes every workgroup slot it finds and keeps it until its
# own queue is empty (profiles/r05/c4_ne4_assembly_vs_hipcc.txt) — one queue can: a wavefront that finishes a task of one group takes the
# next task whatever group it belongs to.  Kernel arguments: a header (queue-level values) + one phf_hier3_isa_args block per body.
BODY_PRIORITY = {3: 0, 4: 2, 5: 3, 6: 3}
FUSED_HDR = {"consts": 0, "queue": 8, "scratch": 16, "t_begin": 24, "t_end": 28, "quantum": 32, "num_tasks": 36, "blocks_magic": 40,
             "rows_per_quantum": 44, "total_blocks": 48, "bounds": 64, "prior_loc": 128, "prior_inv_scale": 168, "prior_shape_m1": 208}
FUSED_HDR_BYTES, FUSED_MAX_BODIES = 256, 16
FUSED_BODY_BYTES = ARG_OFF["prior_loc"]             # a body's block: phf_hier3_isa_args up to the prior (the prior is the run's: once, in the header)


class FusedMain(Main):
    def __init__(self, kinds):
        global WAVE_LDS
        assert len(kinds) <= FUSED_MAX_BODIES
        self.kinds = list(kinds)
        wave_lds, slots = [], set()
        for ne, shape in self.kinds:
            configure(ne, shape)
            wave_lds.append(WAVE_LDS)
            slots.add(NSLOTS)
        assert len(slots) == 1, "the bodies share the wavefront's LDS base arithmetic"
        biggest = None
        for ne, shape in self.kinds:                                           # the prologue's scratch base: a wavefront's area holds the largest tier
            configure(ne, shape)
            if biggest is None or len(GLB_L) > biggest[0]:
                biggest = (len(GLB_L), (ne, shape))
        configure(*biggest[1])
        WAVE_LDS = max(wave_lds)
        self.lds_bytes = WAVE_BASE + 4 * WAVE_LDS
        assert 2 * self.lds_bytes <= 160 * 1024
        self.g = Gen("phf_hier_fused_advance", NSLOTS)
        self.k, self.m, self.c = self.g.k, self.g.m, self.g.c
        self.info = {}
        self.ARG_OFF = self.offsets(0)
        self.emit_end = False
        self.write_through_release = self.WRITE_THROUGH_RELEASE

    def offsets(self, body):
        off = {name: FUSED_HDR_BYTES + body * FUSED_BODY_BYTES + o for name, o in ARG_OFF.items() if o < FUSED_BODY_BYTES}
        off.update({name: FUSED_HDR[name] for name in ("consts", "queue", "scratch", "prior_loc", "prior_inv_scale", "prior_shape_m1")})
        return off

    def next_task_fused(self):
        """pull task n from queue[0]: quantum n / blocks of block n % blocks (all groups' blocks numbered through, group after group);
        wait for that block's previous quantum; then to the body of the block's group (bounds[b] = first block of body b's group)"""
        k, g = self.k, self.g
        H = FUSED_HDR
        k.comment("---- next task (any group's) ----")
        self.l_task, self.l_end = k.new_label("task"), k.new_label("end")
        self.s_qblock = self.s_wave                          # (the wavefront's number in its workgroup is spent after the prologue)
        k.label(self.l_task)
        k.raw_rec("s_setprio 0")                             # (between tasks — pulling, polling — a wavefront does not compete with its partner)
        k.sop("s_mov_b64", EXEC, -1)
        q4 = k.sx(4)
        k.s_load(q4, g.kernarg, H["quantum"])                # quantum num_tasks blocks_magic rows_per_quantum
        a_quant, a_ntasks, a_bmagic, a_rpq = (q4.sub(i, 1) for i in range(4))
        tb = k.sd()
        k.s_load(tb, g.kernarg, H["t_begin"])                # t_begin t_end
        tot = k.s1()
        k.s_load(tot, g.kernarg, H["total_blocks"])
        tmp, task = k.s1(), k.s1()
        vt, one, save = k.v1(), k.v1(), k.sd()
        k.mov32(one, 1)
        k.emit("v_cmp_eq_u32_e32", [VCC], [0, self.v_lane], "valu", count="valu_int")
        k.sop("s_and_saveexec_b64", save, VCC)
        k.emit("global_atomic_add", [vt], [self.v_zero, one, self.s_queue], "vmem", suffix="sc0", mem="vm")
        k.sop("s_mov_b64", EXEC, save)
        k.readfirstlane(task, vt)
        k.sop("s_cmp_ge_u32", None, task, a_ntasks)
        k.branch("s_cbranch_scc1", self.l_end)
        self.udiv(self.s_quant, self.s_qblock, task, tot, a_bmagic, tmp)
        cur, polls = k.sd(), k.s1()
        l_poll, l_ready, l_giveup = k.new_label("poll"), k.new_label("ready"), k.new_label("giveup")
        k.sop("s_lshl_b32", cur.lo(), self.s_qblock, 2)
        k.sop("s_mov_b32", cur.hi(), 0)
        self.add64(cur, cur, self.s_queue)
        k.sop("s_mov_b32", polls, 0)
        k.sop("s_cmp_eq_u32", None, self.s_quant, 0)
        k.branch("s_cbranch_scc1", l_ready)
        k.label(l_poll)
        k.emit("global_load_dword", [vt], [self.v_zero, cur], "vmem", suffix="offset:4 sc1", mem="vm")
        k.readfirstlane(tmp, vt)
        k.sop("s_cmp_ge_i32", None, tmp, self.s_quant)
        k.branch("s_cbranch_scc1", l_ready)
        k.raw_rec("s_sleep 16")
        k.sop("s_add_u32", polls, polls, 1)
        k.sop("s_cmp_lt_u32", None, polls, Lit(1 << 22))
        k.branch("s_cbranch_scc1", l_poll)
        k.label(l_giveup)                                    # cannot happen in a correct run: poison the counter, raise the sticky fault word
        k.sop("s_lshl_b32", cur.lo(), tot, 2)
        k.sop("s_mov_b32", cur.hi(), 0)
        self.add64(cur, cur, self.s_queue)
        k.mov32(one, 1)
        k.emit("global_store_dword", [], [self.v_zero, one, cur], "vmem", suffix="offset:4 sc1", mem="vm")
        k.mov32(one, Lit(0x40000000))
        k.emit("global_store_dword", [], [self.v_zero, one, self.s_queue], "vmem", suffix="sc1", mem="vm")
        k.branch("s_branch", self.l_end)
        k.label(l_ready)
        k.raw_rec("buffer_inv sc1")
        k.free(cur, polls, vt, one, save)
        k.sop("s_mul_i32", tmp, self.s_quant, a_quant)        # this quantum's iterations: (t_begin + quantum * Q, min(that + Q, t_end)]
        k.sop("s_add_u32", self.s_first, tb.lo(), tmp)
        k.sop("s_add_u32", self.s_tend, self.s_first, a_quant)
        k.sop("s_min_u32", self.s_tend, self.s_tend, tb.hi())
        k.sop("s_add_u32", self.s_first, self.s_first, 1)
        k.sop("s_mul_i32", task, self.s_quant, a_rpq)         # rows saved by this block's earlier quanta
        k.sop("s_mov_b32", self.s_t, self.s_first)
        k.free(q4, tb, tot)
        n = len(self.kinds)
        bq = k.sx(4)                                          # four bounds at a time, from the last bodies' down (scalar registers are scarce here)
        self.l_body = [k.new_label("body%d" % b) for b in range(n)]
        picks = [k.new_label("pick%d" % b) for b in range(n)]
        for q in range((n - 1) // 4, -1, -1):
            k.s_load(bq, g.kernarg, H["bounds"] + 16 * q)
            for b in range(min(4 * q + 3, n - 1), 4 * q - 1, -1):
                if b == 0:
                    continue                                  # (falls through to body 0's pick: bounds[0] = 0)
                k.sop("s_cmp_ge_u32", None, self.s_qblock, bq.sub(b % 4, 1))
                k.branch("s_cbranch_scc1", picks[b])
        jt = k.sd()
        for b in range(n):
            if b:
                k.label(picks[b])                             # (reached from its quad's compares: bq holds that quad)
            k.sop("s_sub_u32", self.s_block, self.s_qblock, bq.sub(b % 4, 1))
            k.long_jump(self.l_body[b], jt)
        k.free(bq, jt)
        k.label(self.l_end)                                  # (here, within reach of the task code's branches; the bodies come back by long jumps)
        k.endpgm()
        return tmp, task

    def build(self):
        k = self.k
        self.prologue_common()
        tmp, task = self.next_task_fused()
        snap = [(set(pl.free), pl) for pl in (k.v, k.s)]
        for b, (ne, shape) in enumerate(self.kinds):
            configure(ne, shape)
            for free, pl in snap:
                pl.free = set(free)
            self.ARG_OFF = self.offsets(b)
            k.comment("======== body %d: %d experiments, %s points ========" % (b, ne, " + ".join(str(x) for x in shape)))
            k.label(self.l_body[b])
            # a block's iterations are a serial chain, and a step cannot end before the longest chain does: the bodies whose iteration takes
            # longest get the SIMD's instruction arbiter ahead of their partner wavefront (three-experiment blocks are throughput, not chain)
            k.raw_rec("s_setprio %d" % BODY_PRIORITY.get(ne, 0))
            self.alloc_state()
            w4 = k.sx(4)
            k.s_load(w4, self.g.kernarg, self.ARG_OFF["num_problems"])     # num_problems bpp bpp_magic total_waves
            self.block_setup(w4, tmp, task)
            self.emit_loop()
        configure(3, (4, 4, 4))
        lines_meta = k.finish(self.lds_bytes, FUSED_HDR_BYTES + len(self.kinds) * FUSED_BODY_BYTES)
        return lines_meta, {"vgpr_high_water": k.v.high, "sgpr_high_water": k.s.high, "lds_bytes_per_workgroup": self.lds_bytes,
                            "bodies": len(self.kinds)}


def fused_kernel():
    try:
        return FusedMain(HIER_KERNELS).build()
    finally:
        configure(3, (4, 4, 4))


# the kernels of the code object: (experiments, points per experiment).  The Crumb set's 154 pairs with three experiments are 147 x (4, 4, 4),
# 6 x (2, 2, 2) and 1 x (5, 5, 4)
# ... and the 41 with four are 32 x (4, 4, 4, 1), 5 x (4, 4, 4, 2), 2 x (4, 4, 4, 3), (2, 2, 2, 1), (5, 5, 5, 1)
# ... the 12 with five: 5 x (4, 4, 4, 1, 1), 5 x (4, 4, 4, 2, 1), (4, 4, 4, 4, 4), (5, 5, 4, 2, 2); the 3 with six: 2 x (4, 4, 4, 1, 1, 1), (4, 4, 4, 4, 2, 1)
# In the order of the fused kernel's bodies = the order of a run's block numbering: the bodies whose blocks take longest FIRST, so that in every
# round of the queue their tasks are pulled first (a block's quanta are a serial chain: the long chains must not also start late)
HIER_KERNELS = [(6, (4, 4, 4, 4, 2, 1)), (6, (4, 4, 4, 1, 1, 1)), (5, (4, 4, 4, 4, 4)), (5, (5, 5, 4, 2, 2)), (5, (4, 4, 4, 2, 1)), (5, (4, 4, 4, 1, 1)),
                (4, (5, 5, 5, 1)), (4, (4, 4, 4, 3)), (4, (4, 4, 4, 2)), (4, (4, 4, 4, 1)), (4, (2, 2, 2, 1)),
                (3, (5, 5, 4)), (3, (4, 4, 4)), (3, (2, 2, 2))]
# (The bodies for six experiments first made a C4 step LONGER — 42.4 -> 47.2 ms: a block's 2 000 iterations are a serial chain, and at ~23 us per
# iteration beside a second wavefront that chain outlasted the rest of the grid's work — until the long bodies got the instruction arbiter's
# priority (BODY_PRIORITY) and the first places in the block numbering: 39 ms, and no kernel left beside the grid.  profiles/r05/c4_fused_launch.txt.)


def main_kernel(ne=3, shape=(4, 4, 4)):
    configure(ne, shape)
    try:
        return Main().build()
    finally:
        configure(3, (4, 4, 4))


# ... of which these also exist as kernels of their own (a launch group on its own: phf_hierarchical_advance_queued; s3h; the A/B and PMC runs);
# the others are bodies of the fused kernel only — a group of such a shape launched by itself runs the hipcc kernel (the same numbers)
STANDALONE = [(3, (4, 4, 4)), (3, (2, 2, 2)), (3, (5, 5, 4)), (4, (4, 4, 4, 1)), (4, (4, 4, 4, 2)), (4, (4, 4, 4, 3))]


def main_kernels():
    """[(n_expts, shape, shape code, kernel name, (lines, meta) or None, info)]: every body of the fused kernel, with the kernel of its own
    where it has one"""
    out = []
    for ne, shape in HIER_KERNELS:
        built, info = main_kernel(ne, shape)
        out.append((ne, shape, shape_code(shape), kernel_name(ne, shape), built if (ne, shape) in STANDALONE else None, dict(info)))
    return out


def header_extra(kernels):
    info = [kd for kd in kernels if kd[3] == "phf_hier3_advance"][0][5]
    text = "#define PHF_ISA_HIER3_LDS_BYTES %d\n#define PHF_ISA_HIER3_VGPRS %d\n\n" % (info["lds_bytes_per_workgroup"], info["vgpr_high_water"])
    text += ("/* the hierarchical iteration's kernels: experiments, PHF_HIER_SHAPE code of the point shape (phf_hier_points.points_per_expt),\n"
             " * scratch_slots (doubles per lane of device-memory scratch a resident wavefront keeps part of its chains' state in), name */\n"
             "#define PHF_ISA_HIER_NUM_KERNELS %d\n"
             "static const struct { int n_expts; int shape_code; int scratch_slots; const char* name; } phf_isa_hier_kernels[PHF_ISA_HIER_NUM_KERNELS] = {\n" % len(kernels))
    for ne, shape, code, name, _, kinfo in kernels:
        assert code is not None
        text += "    {%d, %d, %d, \"%s\"}, /* %s */\n" % (ne, code, kinfo["scratch_slots_per_wavefront"], name, " + ".join(str(n) for n in shape))
    text += "};\n\n"
    n = len(kernels)
    from gen_hier_isa import ARGS
    body_fields = [(name, ty) for name, ty in ARGS if ARG_OFF[name] < FUSED_BODY_BYTES]
    text += ("/* phf_hier_fused_advance: ONE persistent grid for every launch group of a run — a header + one argument block per body; body b is\n"
             " * phf_isa_hier_kernels[b]'s iteration, bounds[b] the first block of its group in the run's numbering (blocks of a body without a\n"
             " * group: bounds[b] = bounds[b + 1]; beyond the last body: total_blocks) */\n"
             "#define PHF_ISA_FUSED_BODIES %d\n"
             "typedef struct phf_hier_body_args {\n%s} phf_hier_body_args;    /* phf_hier3_isa_args up to the prior */\n"
             "typedef struct phf_hier_fused_args {\n"
             "  const void* consts;\n  int32_t* queue;\n  double* scratch;\n  uint32_t t_begin, t_end;\n"
             "  uint32_t quantum, num_tasks, blocks_magic, rows_per_quantum;\n  int32_t total_blocks;\n  int32_t pad[3];\n  uint32_t bounds[16];\n"
             "  double prior_loc[5], prior_inv_scale[5], prior_shape_m1[5];\n  double pad1;\n"
             "  phf_hier_body_args body[PHF_ISA_FUSED_BODIES];\n} phf_hier_fused_args;\n"
             % (n, "".join("  %s %s;\n" % (ty, name) for name, ty in body_fields)))
    for name in ("consts", "queue", "scratch", "t_begin", "t_end", "quantum", "num_tasks", "blocks_magic", "rows_per_quantum", "total_blocks", "bounds",
                 "prior_loc", "prior_inv_scale", "prior_shape_m1"):
        text += "_Static_assert(__builtin_offsetof(phf_hier_fused_args, %s) == %d, \"layout of %s\");\n" % (name, FUSED_HDR[name], name)
    text += ("_Static_assert(sizeof(phf_hier_body_args) == %d && __builtin_offsetof(phf_hier3_isa_args, prior_loc) == %d, \"a body's block\");\n"
             "_Static_assert(__builtin_offsetof(phf_hier_fused_args, body) == %d && sizeof(phf_hier_fused_args) == %d, \"layout of the bodies' blocks\");\n\n"
             % (FUSED_BODY_BYTES, FUSED_BODY_BYTES, FUSED_HDR_BYTES, FUSED_HDR_BYTES + n * FUSED_BODY_BYTES))
    return text
