#!/usr/bin/env python3
"""Emit the hand-allocated gfx950 code object source: the hierarchical sampler's one-lane iteration per (experiments, point shape) — Ne = 3 and
Ne = 4, every shape of the Crumb set — and ONE persistent grid with a body per shape (tools/gen_hier_isa_main.py); the single-level model-2
iteration (tools/gen_sl_isa_main.py); unit kernels of every elementary function.  What follows describes the first of them, Ne = 3 with
4 + 4 + 4 points; the others differ in the target phase (emitted per shape) and, for Ne = 4, in a third state tier (device-memory scratch).

    python tools/gen_hier_isa.py            writes pyhillfit_amd/csrc/generated/phf_hier3_gfx950.s and phf_hier3_isa_layout.h
    python tools/gen_hier_isa.py --check    exits 1 if the committed files differ from what this script emits (tests/test_isa_generator.py)
    python tools/gen_hier_isa.py --stats    instruction counts of the loop body and the register budget

WHY (profiles/r04/c4_experiments.txt): the hipcc build of hier_advance_kernel<3> needs all 512 registers (256
VGPR + 256 AGPR, 585 v_accvgpr moves and 852 B/lane of scratch per iteration), so one wavefront per SIMD, 73 % of its issue slots used.
The operation sequence of an iteration is fixed (it has to be: the twin in oracle/phf_oracle.c replays it bit for bit), so it can be
EMITTED: this script walks the same sequence as phf_hier_model.h / phf_hierarchical.hip (every fp64 operation in the same order with
the same operands — tools/isa/phf_isa_math.py for phf_math.h) and places every value itself:

  VGPRs (256, two wavefronts per SIMD)   theta, log target, the proposal, the register part of L, loga, accepted count, the scale,
                                         log u; three polynomial constants; everything else is a temporary with an explicit lifetime
  LDS   (per wavefront, [slot][64 lanes]) the running mean, the diagonal d and LDS_L elements of L — each read once and written once
                                         per iteration, by the sweep
  LDS   (per 256-thread workgroup)       ONE copy of the exp2 / log / erfc / normal tables (8 KB) for its four wavefronts
  SGPRs                                  constants of the polynomials, loop state, the Philox key schedule, the points (scalar loads)

The loop is rotated against the C source so that one copy of every phase serves the whole launch:
    for t = t_begin + 1 .. t_end + 1:   draws(t) -> sweep(adaptation of t - 1; g = 0 in the first pass: an exact no-op that still
                                        yields y = L sqrt(d) z) -> save(t - 1) -> [t > t_end: leave] -> propose -> target -> accept
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "isa"))
import gfx950_asm as A                     # noqa: E402
import phf_isa_math as M                   # noqa: E402
from gfx950_asm import EXEC, VCC, Lit, Neg, Reg  # noqa: E402

OUT_DIR = os.path.join(HERE, "..", "pyhillfit_amd", "csrc", "generated")
CONST_OFF = 8064                            # byte offset of the scalar constants inside the blob (tables first, padded to 64)

NE, D = 3, 11
TRI = D * (D + 1) // 2
S_TH, S_LT, S_MEAN, S_TRI, S_LOGA, S_NACC = 0, D, D + 1, 2 * D + 1, 2 * D + 1 + TRI, 2 * D + 2 + TRI       # state rows (phf_hierarchical.hip)

# kernel arguments of phf_hier3_advance (bytes); the C struct is generated from this list (phf_hier3_isa_layout.h)
ARGS = [("consts", "const void*"), ("state", "double*"), ("rows", "double*"), ("moments", "double*"), ("gamma", "const double*"),
        ("ln_conc", "const double*"), ("response", "const double*"), ("pair_index", "const int32_t*"), ("problem_id", "const uint32_t*"),
        ("launch_order", "const int32_t*"), ("chain_offset", "const uint32_t*"), ("queue", "int32_t*"),
        ("t_begin", "uint32_t"), ("t_end", "uint32_t"), ("adapt_start", "uint32_t"), ("thinning", "int32_t"),
        ("moments_after", "uint32_t"), ("chains", "int32_t"), ("num_problems", "int32_t"), ("bpp", "int32_t"),
        ("bpp_magic", "uint32_t"), ("total_waves", "int32_t"), ("seed_lo", "uint32_t"), ("seed_hi", "uint32_t"),
        ("chain_id_base", "uint32_t"), ("pts_stride", "int32_t"), ("until_save0", "int32_t"), ("quantum", "uint32_t"),
        ("num_tasks", "int32_t"), ("blocks_magic", "uint32_t"), ("rows_per_quantum", "uint32_t"), ("pad0", "int32_t"),
        ("prior_loc", "double[5]"), ("prior_inv_scale", "double[5]"), ("prior_shape_m1", "double[5]"), ("three_twelve", "double[2]"),
        ("scratch", "double*")]


def arg_layout(args=None):
    off, out = 0, {}
    for name, ty in (ARGS if args is None else args):
        if ty.endswith("*"):
            size = 8
        elif ty.startswith("double["):
            size = 8 * int(ty[7:-1])
        else:
            size = 4
        al = 8 if size >= 8 else 4
        off = (off + al - 1) // al * al
        out[name] = off
        off += size
    return out, (off + 7) // 8 * 8


ARG_OFF, ARG_BYTES = arg_layout()

# kernel arguments of phf_sl3_advance (single-level model 2).  The fields the shared task code reads (tools/gen_hier_isa_main.py:
# task_fetch / task_done) carry the same names and the same order within their 16-byte groups as in ARGS.
SL_ARGS = [("consts", "const void*"), ("state", "double*"), ("rows", "double*"), ("gamma", "const double*"),
           ("ln_conc", "const double*"), ("response", "const double*"), ("weight", "const double*"), ("counts", "const int32_t*"),
           ("pi_bit", "const double*"), ("extra", "const double*"), ("pair_index", "const int32_t*"), ("temperature", "const double*"),
           ("problem_id", "const uint32_t*"), ("launch_order", "const int32_t*"), ("chain_offset", "const uint32_t*"), ("queue", "int32_t*"),
           ("t_begin", "uint32_t"), ("t_end", "uint32_t"), ("adapt_start", "uint32_t"), ("thinning", "int32_t"),
           ("reset_mean", "uint32_t"), ("chains", "int32_t"), ("num_problems", "int32_t"), ("bpp", "int32_t"),
           ("bpp_magic", "uint32_t"), ("total_waves", "int32_t"), ("seed_lo", "uint32_t"), ("seed_hi", "uint32_t"),
           ("chain_id_base", "uint32_t"), ("pts_stride", "int32_t"), ("until_save0", "int32_t"), ("quantum", "uint32_t"),
           ("num_tasks", "int32_t"), ("blocks_magic", "uint32_t"), ("rows_per_quantum", "uint32_t"), ("pad0", "int32_t")]
SL_ARG_OFF, SL_ARG_BYTES = arg_layout(SL_ARGS)


class Gen(object):
    """state shared by the phases of one kernel"""

    def __init__(self, name, wave_lds_slots=0):
        self.k = A.Kernel(name, num_vgpr=256, num_sgpr=102, first_sgpr=3, first_vgpr=1)
        self.kernarg = Reg("s", 0, 2)
        self.wg_id = Reg("s", 2, 1)
        self.tid = Reg("v", 0, 1)
        self.c = {}
        self.m = M.Ctx(self.k, self.c)
        self.slots = wave_lds_slots

    # the resident constants: SGPR pairs loaded from the blob, three of them copied to VGPRs (operands of an fma that already has one
    # scalar operand: VOP3 reads one constant-bus value)
    def load_constants(self, s_consts, names):
        k = self.k
        for n in names:
            r = k.sd()
            k.s_load(r, s_consts, CONST_OFF + 8 * M.const_index(n))
            self.c[n] = r
        for n in ("MAGIC", "KE2", "KL3"):
            v = k.vd()
            tmp = k.sd()
            k.s_load(tmp, s_consts, CONST_OFF + 8 * M.const_index(n))
            k.emit("v_mov_b32_e32", [v.lo()], [tmp.lo()], "valu", count="valu_int")
            k.emit("v_mov_b32_e32", [v.hi()], [tmp.hi()], "valu", count="valu_int")
            self.c[n + "V"] = v
            k.free(tmp)
        for n, val in (("NINFHI", 0xfff00000), ("ONEHI", 0x3ff00000)):     # high words of -inf and 1.0: selected by v_cndmask (whose mask
            v = k.v1()                                                      # is its one constant-bus read)
            k.mov32(v, Lit(val))
            self.c[n] = v
        for n, val in (("PM0", M.PHILOX_M0), ("PM1", M.PHILOX_M1), ("ABSMASK", 0x7fffffff), ("MANTHI", 0x000fffff), ("NORMBIAS", -48 * (0x3ff << 1))):
            r = k.s1()
            k.sop("s_mov_b32", r, Lit(val))
            self.c[n] = r

    def stage_tables(self, s_consts):
        """the 8 048 bytes of tables: blob -> LDS offset 0, by all 256 threads (thread i: doubles i, i + 256, ...); then the barrier.
        The last round is partial: lanes beyond the tables load (harmlessly) the padding / constants behind them and do not store."""
        k = self.k
        rounds = (M.TABLE_BYTES // 8 + 255) // 256
        off = [k.v1() for _ in range(rounds)]
        tmp = [k.vd() for _ in range(rounds)]
        k.vop("v_lshlrev_b32_e32", off[0], 3, self.tid)
        for j in range(1, rounds):
            k.vop("v_add_u32_e32", off[j], Lit(2048 * j), off[0])
        for j in range(rounds):
            k.gload(tmp[j], off[j], s_consts, 0)
        for j in range(rounds):
            if (j + 1) * 2048 > M.TABLE_BYTES:
                save = k.sd()
                k.emit("v_cmp_gt_u32_e32", [VCC], [Lit(M.TABLE_BYTES), off[j]], "valu", count="valu_int")
                k.sop("s_and_saveexec_b64", save, VCC)
                k.ds_write(off[j], tmp[j], 0)
                k.sop("s_mov_b64", EXEC, save)
                k.free(save)
            else:
                k.ds_write(off[j], tmp[j], 0)
        k.free(off, tmp)
        k.barrier()


    def stage_ranges(self, s_consts, ranges):
        """ranges: (byte offset in the blob, byte offset in LDS, bytes) — copied by all 256 threads, 8 bytes per thread and round; then
        the workgroup's barrier"""
        k = self.k
        t8 = k.v1()
        k.vop("v_lshlrev_b32_e32", t8, 3, self.tid)
        for blob_off, lds_off, nbytes in ranges:
            rounds = (nbytes // 8 + 255) // 256
            for r0 in range(0, rounds, 4):
                batch = list(range(r0, min(r0 + 4, rounds)))
                offs = [k.v1() for _ in batch]
                tmp = [k.vd() for _ in batch]
                for i, j in enumerate(batch):
                    k.vop("v_add_u32_e32", offs[i], Lit(2048 * j), t8)
                for i, j in enumerate(batch):
                    sb = k.sd()
                    k.sop("s_add_u32", sb.lo(), s_consts.lo(), Lit(blob_off))
                    k.sop("s_addc_u32", sb.hi(), s_consts.hi(), 0)
                    k.gload(tmp[i], offs[i], sb)
                    k.free(sb)
                for i, j in enumerate(batch):
                    if (j + 1) * 2048 > nbytes:
                        save = k.sd()
                        k.emit("v_cmp_gt_u32_e32", [VCC], [Lit(nbytes), offs[i]], "valu", count="valu_int")
                        k.sop("s_and_saveexec_b64", save, VCC)
                        k.ds_write(offs[i], tmp[i], lds_off)
                        k.sop("s_mov_b64", EXEC, save)
                        k.free(save)
                    else:
                        k.ds_write(offs[i], tmp[i], lds_off)
                k.free(offs, tmp)
        k.free(t8)
        k.barrier()


ALL_CONSTS = [n for n, _ in M.CONSTS if n not in ("MAGIC", "KE2", "KL3")]


# ---------------------------------------------------------------------------------------------------------------- unit kernels
def unit_kernel(name, body, in_kind="f64", out_words=2):
    """y[i] = f(x[i]): kernel arguments {consts, in, out, n (u32)}; one element per thread, 256 threads per workgroup.
    body(g, x) -> result register(s).  Test infrastructure for tests/test_gpu_isa.py (each function against the host build of
    phf_math.h), shipped inside the same code object."""
    g = Gen(name)
    k = g.k
    a = k.sx(8)
    k.s_load(a, g.kernarg, 0)                          # consts, in, out, n
    s_consts, s_in, s_out, s_n = a.sub(0), a.sub(2), a.sub(4), a.sub(6, 1)
    g.load_constants(s_consts, ALL_CONSTS)
    g.stage_tables(s_consts)
    gid = k.v1()
    k.vop("v_lshl_or_b32", gid, g.wg_id, 8, g.tid)
    k.cmp_u32("gt", VCC, s_n, gid)
    k.sop("s_and_b64", EXEC, EXEC, VCC)
    done = k.new_label("done")
    k.branch("s_cbranch_execz", done)
    off = k.v1()
    if in_kind == "f64":
        x = k.vd()
        k.vop("v_lshlrev_b32_e32", off, 3, gid)
        k.gload(x, off, s_in)
    elif in_kind == "u32":
        x = k.v1()
        k.vop("v_lshlrev_b32_e32", off, 2, gid)
        k.gload(x, off, s_in)
    else:                                               # u32 x 6: Philox counter and key
        x = [k.v1() for _ in range(6)]
        k.vop("v_mul_u32_u24_e32", off, 24, gid)
        for j in range(6):
            k.gload(x[j], off, s_in, 4 * j)
    res = body(g, x)
    if out_words == 2:
        k.vop("v_lshlrev_b32_e32", off, 3, gid)
        k.gstore(off, res, s_out)
    else:
        k.vop("v_lshlrev_b32_e32", off, 4, gid)
        for j in range(4):
            k.gstore(off, res[j], s_out, 4 * j)
    k.label(done)
    k.endpgm()
    return k.finish(M.TABLE_BYTES + 16, 32)


def unit_kernels():
    out = []

    def one(fn):
        def body(g, x):
            d = g.k.vd()
            fn(g.m, [d], [x])
            return d
        return body

    def exp_fast_body(g, x):
        d = g.k.vd()
        M.exp_fast(g.m, [d], [x])
        return d

    def exp_capped_body(g, x):
        d = g.k.vd()
        M.exp_capped(g.m, [d], [x])
        return d

    def sqrt_body(g, x):
        d, mk = g.k.vd(), g.k.sd()
        M.sqrt_nonneg(g.m, [d], [x], [mk])
        return d

    def normal_body(g, w):
        d = g.k.vd()
        M.normal_u32(g.m, [d], [w])
        return d

    def logu_body(g, w):
        d = g.k.vd()
        M.unit_open32(g.m, d, w)
        M.log_pos(g.m, [d], [d])
        return d

    def philox_body(g, x):
        k = g.k
        # the key schedule as the kernels do it: on the scalar unit, from the (uniform) seed — here the key is per element, so read
        # it through v_readfirstlane is not possible: the unit test uses ONE key for the whole launch (element 0's)
        k0, k1 = k.s1(), k.s1()
        k.readfirstlane(k0, x[4])
        k.readfirstlane(k1, x[5])
        keys = []
        for r in range(7):
            a, b = k.s1(), k.s1()
            k.sop("s_add_u32", a, k0, Lit((r * M.PHILOX_W0) & 0xffffffff))
            k.sop("s_add_u32", b, k1, Lit((r * M.PHILOX_W1) & 0xffffffff))
            keys.append((a, b))
        words, _ = M.philox(g.m, x[0], x[1], x[2], x[3], keys)
        return words

    out.append(unit_kernel("phf_isa_unit_exp_fast", exp_fast_body))
    out.append(unit_kernel("phf_isa_unit_exp_capped", exp_capped_body))
    out.append(unit_kernel("phf_isa_unit_log_pos", one(M.log_pos)))
    out.append(unit_kernel("phf_isa_unit_log_fast", one(M.log_fast)))
    out.append(unit_kernel("phf_isa_unit_erfc_tab", one(M.erfc_tab)))
    out.append(unit_kernel("phf_isa_unit_rcp", one(M.rcp)))
    out.append(unit_kernel("phf_isa_unit_sqrt_nonneg", sqrt_body))
    out.append(unit_kernel("phf_isa_unit_normal_u32", normal_body, in_kind="u32"))
    out.append(unit_kernel("phf_isa_unit_log_u", logu_body, in_kind="u32"))
    out.append(unit_kernel("phf_isa_unit_philox7", philox_body, in_kind="u32x6", out_words=4))
    return out


def layout_header():
    lines = ["/* GENERATED by tools/gen_hier_isa.py — do not edit.  Layout shared by the gfx950 assembly (phf_hier3_gfx950.s) and its host side",
             " * (phf_hier3_isa.hip): the kernel-argument block, the constants blob, the LDS budget. */",
             "#ifndef PHF_HIER3_ISA_LAYOUT_H", "#define PHF_HIER3_ISA_LAYOUT_H", "#include <stdint.h>", "",
             "#define PHF_ISA_TABLE_BYTES %d" % M.TABLE_BYTES, "#define PHF_ISA_CONST_OFF %d" % CONST_OFF,
             "#define PHF_ISA_EXP2_OFF %d" % M.EXP2_OFF, "#define PHF_ISA_LOG_OFF %d" % M.LOG_OFF,
             "#define PHF_ISA_ERFC_OFF %d" % M.ERFC_OFF, "#define PHF_ISA_NORMAL_OFF %d" % M.NORMAL_OFF,
             "#define PHF_ISA_NUM_CONSTS %d" % len(M.CONSTS),
             "/* bit patterns of the scalar constants, in blob order (names: tools/isa/phf_isa_math.py CONSTS) */",
             "static const uint64_t phf_isa_const_bits[PHF_ISA_NUM_CONSTS] = {"]
    for n, v in M.CONSTS:
        bits = M.CONST_BITS[n] if v is None else A.f64_bits(v)
        lines.append("    0x%016xull, /* %s */" % (bits, n))
    lines += ["};", "", "typedef struct phf_hier3_isa_args {"]
    for name, ty in ARGS:
        if ty.startswith("double["):
            lines.append("  double %s[%s];" % (name, ty[7:-1]))
        else:
            lines.append("  %s %s;" % (ty, name))
    lines += ["} phf_hier3_isa_args;", ""]
    for name, _ in ARGS:
        lines.append("_Static_assert(__builtin_offsetof(phf_hier3_isa_args, %s) == %d, \"layout of %s\");" % (name, ARG_OFF[name], name))
    lines += ["_Static_assert(sizeof(phf_hier3_isa_args) == %d, \"size of the argument block\");" % ARG_BYTES, "",
              "typedef struct phf_sl3_isa_args {"]
    for name, ty in SL_ARGS:
        lines.append("  %s %s;" % (ty, name))
    lines += ["} phf_sl3_isa_args;", ""]
    for name, _ in SL_ARGS:
        lines.append("_Static_assert(__builtin_offsetof(phf_sl3_isa_args, %s) == %d, \"layout of %s\");" % (name, SL_ARG_OFF[name], name))
    lines += ["_Static_assert(sizeof(phf_sl3_isa_args) == %d, \"size of the argument block\");" % SL_ARG_BYTES, "",
              "#endif", ""]
    return "\n".join(lines)


def generate(with_main=True):
    kernels = unit_kernels()
    info = {}
    if with_main:
        import gen_hier_isa_main as G
        mains = G.main_kernels()
        info = dict([kd for kd in mains if kd[3] == "phf_hier3_advance"][0][5])
        for ne, shape, code, name, built, kinfo in mains:
            if built is not None:
                kernels.append(built)
            if name != "phf_hier3_advance":
                info["%s" % name] = {"vgpr_high_water": kinfo["vgpr_high_water"], "lds_bytes_per_workgroup": kinfo["lds_bytes_per_workgroup"],
                                     "count_iteration": kinfo["count_iteration"]}
        fused, finfo = G.fused_kernel()
        kernels.append(fused)
        info["phf_hier_fused_advance"] = finfo
        import gen_sl_isa_main as GS
        sl, sl_info = GS.main_kernel()
        kernels.append(sl)
        info.update(sl_info)
        hdr_extra = G.header_extra(mains) + ("#define PHF_ISA_LOGPHI_BLOB_OFF %d\n#define PHF_ISA_SL_MAX_STRIDE %d\n\n" % (GS.BLOB_LOGPHI_OFF, GS.MAX_STRIDE))
    else:
        hdr_extra = ""
    text = "; GENERATED by tools/gen_hier_isa.py — do not edit (regenerate; tests/test_isa_generator.py checks this file against the script)\n"
    text += A.module_text(kernels)
    hdr = layout_header().replace("#endif\n", hdr_extra + "#endif\n")
    return text, hdr, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--units-only", action="store_true")
    a = ap.parse_args()
    text, hdr, info = generate(not a.units_only)
    paths = [(os.path.join(OUT_DIR, "phf_hier3_gfx950.s"), text), (os.path.join(OUT_DIR, "phf_hier3_isa_layout.h"), hdr)]
    if a.check:
        bad = [p for p, t in paths if not os.path.exists(p) or open(p).read() != t]
        if bad:
            print("out of date: " + ", ".join(os.path.relpath(p) for p in bad))
            sys.exit(1)
        print("generated files are current")
        return
    os.makedirs(OUT_DIR, exist_ok=True)
    for p, t in paths:
        with open(p, "w") as f:
            f.write(t)
        print("wrote %s (%d lines)" % (os.path.relpath(p), t.count("\n")))
    if a.stats and info:
        for key in sorted(info):
            print("%-28s %s" % (key, info[key]))


if __name__ == "__main__":
    main()
