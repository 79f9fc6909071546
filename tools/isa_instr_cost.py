#!/usr/bin/env python3
"""Issue cost of single gfx950 vector instructions, measured: one kernel per instruction, each a loop over 64 independent copies (distinct
destination registers, fixed sources), run with 1 / 2 / 4 wavefronts per SIMD on the whole chip; cycles per instruction and wavefront
relative to v_fma_f64 at the same occupancy (taken as 4 cycles: 16 fp64 lanes per SIMD).  Behind the instruction choices of
tools/isa/phf_isa_math.py and the reading of the C3 counters (profiles/r05/gfx950_instruction_costs.txt).

  python tools/isa_instr_cost.py --emit DIR     # here: writes DIR/instr_cost.s and assembles DIR/instr_cost.co (no GPU needed)
  python tools/isa_instr_cost.py --run DIR      # on the GPU box: loads DIR/instr_cost.co through the HIP module API (ctypes), times, prints
"""
import argparse
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa"))
import gfx950_asm as ga  # noqa: E402

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
COPIES = 64

# name -> text of one copy; {d} = destination VGPR index (even), sources are v[100:107] (finite doubles / small integers) and s[8:11]
INSTR = [
    ("v_fma_f64", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]"),
    ("v_mul_f64", "v_mul_f64 v[{d}:{d1}], v[100:101], v[102:103]"),
    ("v_add_f64", "v_add_f64 v[{d}:{d1}], v[100:101], v[102:103]"),
    ("v_max_f64", "v_max_f64 v[{d}:{d1}], v[100:101], v[102:103]"),
    ("v_ldexp_f64", "v_ldexp_f64 v[{d}:{d1}], v[100:101], v106"),
    ("v_rcp_f64", "v_rcp_f64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_rsq_f64", "v_rsq_f64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_rndne_f64", "v_rndne_f64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_cvt_f64_u32", "v_cvt_f64_u32_e32 v[{d}:{d1}], v106"),
    ("v_cvt_f64_i32", "v_cvt_f64_i32_e32 v[{d}:{d1}], v106"),
    ("v_cvt_i32_f64", "v_cvt_i32_f64_e32 v{d}, v[100:101]"),
    ("v_cmp_lt_f64", "v_cmp_lt_f64_e32 vcc, v[100:101], v[102:103]"),
    ("v_cmp_lt_f64_sgpr", "v_cmp_lt_f64_e64 s[12:13], v[100:101], v[102:103]"),
    ("v_cmp_class_f64", "v_cmp_class_f64_e64 s[12:13], v[100:101], v106"),
    ("v_cmp_lt_u32", "v_cmp_lt_u32_e32 vcc, v106, v107"),
    ("v_mov_b32", "v_mov_b32_e32 v{d}, v106"),
    ("v_mov_b64", "v_mov_b64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_pk_mov_b32", "v_pk_mov_b32 v[{d}:{d1}], v[100:101], v[102:103] op_sel:[0,1]"),
    ("v_add_u32", "v_add_u32_e32 v{d}, v106, v107"),
    ("v_add_co_u32", "v_add_co_u32_e32 v{d}, vcc, v106, v107"),
    ("v_and_b32", "v_and_b32_e32 v{d}, v106, v107"),
    ("v_and_b32_lit", "v_and_b32_e32 v{d}, 0x12345, v107"),
    ("v_and_b32_sgpr", "v_and_b32_e32 v{d}, s8, v107"),
    ("v_or3_b32", "v_or3_b32 v{d}, v106, v107, v100"),
    ("v_bfi_b32", "v_bfi_b32 v{d}, v106, v107, v100"),
    ("v_bfe_u32", "v_bfe_u32 v{d}, v106, 3, 9"),
    ("v_lshlrev_b32", "v_lshlrev_b32_e32 v{d}, 3, v107"),
    ("v_lshl_or_b32", "v_lshl_or_b32 v{d}, v106, 3, v107"),
    ("v_lshl_add_u32", "v_lshl_add_u32 v{d}, v106, 3, v107"),
    ("v_add3_u32", "v_add3_u32 v{d}, v106, v107, v100"),
    ("v_ashrrev_i32", "v_ashrrev_i32_e32 v{d}, 20, v107"),
    ("v_lshlrev_b64", "v_lshlrev_b64 v[{d}:{d1}], 3, v[100:101]"),
    ("v_lshrrev_b64", "v_lshrrev_b64 v[{d}:{d1}], 3, v[100:101]"),
    ("v_lshl_add_u64", "v_lshl_add_u64 v[{d}:{d1}], v[100:101], 3, v[102:103]"),
    ("v_cndmask_b32", "v_cndmask_b32_e32 v{d}, v106, v107, vcc"),
    ("v_cndmask_b32_sgpr", "v_cndmask_b32_e64 v{d}, v106, v107, s[12:13]"),
    ("v_mul_lo_u32", "v_mul_lo_u32 v{d}, v106, v107"),
    ("v_mul_hi_u32", "v_mul_hi_u32 v{d}, v106, v107"),
    ("v_mul_u32_u24", "v_mul_u32_u24_e32 v{d}, v106, v107"),
    ("v_mad_u32_u24", "v_mad_u32_u24 v{d}, v106, v107, v100"),
    ("v_mad_u64_u32", "v_mad_u64_u32 v[{d}:{d1}], s[12:13], v106, v107, v[102:103]"),
    ("v_mad_u64_u32_zero", "v_mad_u64_u32 v[{d}:{d1}], s[12:13], v106, v107, 0"),
    ("v_accvgpr_write", "v_accvgpr_write_b32 a{da}, v106"),
    ("v_accvgpr_read", "v_accvgpr_read_b32 v{d}, a0"),
    ("v_readfirstlane", "v_readfirstlane_b32 s12, v106"),
    ("v_fma_f32", "v_fma_f32 v{d}, v106, v107, v100"),
    ("v_pk_fma_f32", "v_pk_fma_f32 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]"),
    ("s_mov_b32", "s_mov_b32 s12, s8"),
    ("s_nop0", "s_nop 0"),
    ("ds_read_b64", "ds_read_b64 v[{d}:{d1}], v108"),
    ("ds_read_b128", "ds_read_b128 v[{d4}:{d4e}], v109"),
    ("ds_read2_b64", "ds_read2_b64 v[{d4}:{d4e}], v108 offset1:64"),
    ("ds_read_b32", "ds_read_b32 v{d}, v108"),
    ("ds_write_b128", "ds_write_b128 v109, v[100:103]"),
    ("ds_read_b64_uniform", "ds_read_b64 v[{d}:{d1}], v110"),
    ("ds_read_b128_uniform", "ds_read_b128 v[{d4}:{d4e}], v110"),
    ("ds_write_b64", "ds_write_b64 v108, v[100:101]"),
    # the selects: where the mask comes from
    ("cnd_e32_vcc_ones", "v_cndmask_b32_e32 v{d}, v106, v107, vcc", "s_mov_b64 vcc, -1"),
    ("cnd_e32_vcc_mixed", "v_cndmask_b32_e32 v{d}, v106, v107, vcc", "s_mov_b64 vcc, 0x55555555"),
    ("cnd_e64_vcc", "v_cndmask_b32_e64 v{d}, v106, v107, vcc"),
    ("cnd_e32_const", "v_cndmask_b32_e32 v{d}, 0, v107, vcc"),
    ("cmp_vcc_2cnd_e32", "v_cmp_lt_f64_e32 vcc, v[100:101], v[102:103]\n\tv_cndmask_b32_e32 v{d}, v106, v107, vcc\n\tv_cndmask_b32_e32 v{d1}, v106, v107, vcc"),
    ("cmp_sgpr_2cnd_e64", "v_cmp_lt_f64_e64 s[12:13], v[100:101], v[102:103]\n\tv_cndmask_b32_e64 v{d}, v106, v107, s[12:13]\n\tv_cndmask_b32_e64 v{d1}, v106, v107, s[12:13]"),
    ("v_addc_co_u32", "v_addc_co_u32_e32 v{d}, vcc, v106, v107, vcc"),
    ("v_min_f64", "v_min_f64 v[{d}:{d1}], v[100:101], v[102:103]"),
    ("v_fma_f64_sgpr", "v_fma_f64 v[{d}:{d1}], v[100:101], s[8:9], v[104:105]"),
    ("v_fma_f64_neg", "v_fma_f64 v[{d}:{d1}], -v[100:101], v[102:103], v[104:105]"),
    ("v_fma_f64_const", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], 1.0"),
    ("v_mul_f64_const", "v_mul_f64 v[{d}:{d1}], v[100:101], 0.5"),
    ("v_add_f64_abs", "v_add_f64 v[{d}:{d1}], |v[100:101]|, v[102:103]"),
    ("v_frexp_exp_i32_f64", "v_frexp_exp_i32_f64_e32 v{d}, v[100:101]"),
    ("v_frexp_mant_f64", "v_frexp_mant_f64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_floor_f64", "v_floor_f64_e32 v[{d}:{d1}], v[100:101]"),
    ("v_cvt_u32_f64", "v_cvt_u32_f64_e32 v{d}, v[100:101]"),
    ("v_sub_u32", "v_sub_u32_e32 v{d}, v106, v107"),
    ("v_xor_b32", "v_xor_b32_e32 v{d}, v106, v107"),
    ("v_or_b32", "v_or_b32_e32 v{d}, v106, v107"),
    ("v_lshrrev_b32", "v_lshrrev_b32_e32 v{d}, 3, v107"),
    ("v_lshlrev_b32_v", "v_lshlrev_b32_e32 v{d}, v106, v107"),
    ("v_ashrrev_i32_v", "v_ashrrev_i32_e32 v{d}, v106, v107"),
    ("v_add_u32_const", "v_add_u32_e32 v{d}, 3, v107"),
    ("v_add_u32_sgpr", "v_add_u32_e32 v{d}, s8, v107"),
    ("v_mov_b32_const", "v_mov_b32_e32 v{d}, 0"),
    ("v_mov_b32_lit", "v_mov_b32_e32 v{d}, 0x12345"),
    ("v_mov_b32_sgpr", "v_mov_b32_e32 v{d}, s8"),
    ("v_max_u32", "v_max_u32_e32 v{d}, v106, v107"),
    ("v_min_u32", "v_min_u32_e32 v{d}, v106, v107"),
    ("v_and_or_b32", "v_and_or_b32 v{d}, v106, v107, v100"),
    ("v_perm_b32", "v_perm_b32 v{d}, v106, v107, v100"),
    ("v_alignbit_b32", "v_alignbit_b32 v{d}, v106, v107, 9"),
    ("mix_fma_mov", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_mov_b32_e32 v{d}, v106"),
    ("mix_fma_and", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_and_b32_e32 v{d4}, v106, v107"),
    ("mix_fma_ds_read", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tds_read_b64 v[{d4}:{d4b}], v108"),
    ("mix_4fma_ds_read", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tds_read_b64 v[{d4}:{d4b}], v108"),
    ("mix_fma_cnd_e32", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_cndmask_b32_e32 v{d4}, v106, v107, vcc"),
    ("mix_fma_cnd_e64", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_cndmask_b32_e64 v{d4}, v106, v107, s[12:13]"),
    ("mix_3fma_cnd_e32", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_cndmask_b32_e32 v{d4}, v106, v107, vcc"),
    ("mix_cmp_2cnd_e32_4fma", "v_cmp_lt_f64_e32 vcc, v[100:101], v[102:103]\n\tv_cndmask_b32_e32 v{d4}, v106, v107, vcc\n\tv_cndmask_b32_e32 v{d4b}, v106, v107, vcc\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]"),
    ("mix_cmp_2cnd_e64_4fma", "v_cmp_lt_f64_e64 s[12:13], v[100:101], v[102:103]\n\tv_cndmask_b32_e64 v{d4}, v106, v107, s[12:13]\n\tv_cndmask_b32_e64 v{d4b}, v106, v107, s[12:13]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\tv_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]"),
    ("mix_fma_salu", "v_fma_f64 v[{d}:{d1}], v[100:101], v[102:103], v[104:105]\n\ts_mov_b32 s12, s8"),
    # pairs: a dependent chain inside ONE wavefront (latency, visible at one wavefront per SIMD)
    ("dep_fma_f64", "v_fma_f64 v[0:1], v[0:1], v[102:103], v[104:105]"),
    ("dep_add_u32", "v_add_u32_e32 v0, v0, v107"),
    ("dep_mul_then_fma", "v_mul_f64 v[{d}:{d1}], v[100:101], v[102:103]\n\tv_fma_f64 v[{d}:{d1}], v[{d}:{d1}], v[102:103], v[104:105]"),
]


def kernel_text(name, body, setup=None):
    lines = [
        "\ts_load_dword s4, s[0:1], 0x0",
        "\tv_lshlrev_b32_e32 v108, 3, v0",
        "\tv_mov_b32_e32 v100, 0x55555555", "\tv_mov_b32_e32 v101, 0x3ff55555",
        "\tv_mov_b32_e32 v102, 0x9999999a", "\tv_mov_b32_e32 v103, 0x3fe99999",
        "\tv_mov_b32_e32 v104, 0x11111111", "\tv_mov_b32_e32 v105, 0x3fb11111",
        "\tv_mov_b32_e32 v106, 5", "\tv_mov_b32_e32 v107, 0x12345",
        "\tv_mov_b32_e32 v0, 0", "\tv_mov_b32_e32 v1, 0x3ff00000",
        "\tv_lshlrev_b32_e32 v109, 4, v0", "\tv_mov_b32_e32 v110, 0",
        "\ts_mov_b32 s8, 0xffff", "\ts_mov_b32 s9, 0x3fe00000", "\ts_mov_b64 s[12:13], 0", "\ts_mov_b64 vcc, 0",
    ] + (["\t" + setup] if setup else []) + [
        "\ts_waitcnt lgkmcnt(0)",
        ".L%s_loop:" % name,
    ]
    for i in range(COPIES):
        d = 2 + 2 * (i % 48)                      # v2..v97, even
        d4 = 4 + 4 * (i % 23)                     # v4..v95 for the quads
        for ln in body.format(d=d, d1=d + 1, d4=d4, d4e=d4 + 3, d4b=d4 + 1, da=d % 16).split("\n"):
            lines.append("\t" + ln.strip())
        if "ds_" in body and i % 8 == 7:
            lines.append("\ts_waitcnt lgkmcnt(0)")
    lines += [
        "\ts_sub_u32 s4, s4, 1",
        "\ts_cmp_lg_u32 s4, 0",
        "\ts_cbranch_scc1 .L%s_loop" % name,
        "\ts_waitcnt vmcnt(0) lgkmcnt(0)",
        "\ts_endpgm",
    ]
    return lines


class _K(object):
    """just enough of gfx950_asm.Kernel for finish(): fixed register counts, prepared lines"""
    def __init__(self, name, lines):
        self.name, self.lines = name, lines

    def finish(self):
        class P(object):
            pass
        k = ga.Kernel.__new__(ga.Kernel)
        k.name, k.lines = self.name, self.lines
        k.v, k.s = P(), P()
        k.v.high, k.s.high = 112, 24
        k.finalize = lambda: None
        return ga.Kernel.finish(k, 4096, 8)


def emit(out):
    os.makedirs(out, exist_ok=True)
    kernels = [_K("ic_" + t[0], kernel_text("ic_" + t[0], *t[1:])).finish() for t in INSTR]
    # 112 VGPRs + 16 accumulation registers (for the v_accvgpr kernels) = 128: four wavefronts per SIMD fit
    text = ga.module_text(kernels).replace(".agpr_count:     0", ".agpr_count:     16")
    text = text.replace(".amdhsa_next_free_vgpr 112", ".amdhsa_next_free_vgpr 128").replace(".vgpr_count:     112", ".vgpr_count:     128")
    src, obj, co = (os.path.join(out, "instr_cost" + e) for e in (".s", ".o", ".co"))
    open(src, "w").write(text)
    subprocess.check_call([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj])
    subprocess.check_call([os.path.join(LLVM_BIN, "ld.lld"), "-shared", obj, "-o", co])
    print(co)


def run(out, reps):
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    def ck(e, what):
        if e:
            raise RuntimeError("%s: hip error %d" % (what, e))
    ck(hip.hipInit(0), "hipInit")
    ck(hip.hipSetDevice(0), "hipSetDevice")
    blob = open(os.path.join(out, "instr_cost.co"), "rb").read()
    mod = ctypes.c_void_p()
    ck(hip.hipModuleLoadData(ctypes.byref(mod), blob), "hipModuleLoadData")
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    ck(hip.hipEventCreate(ctypes.byref(ev0)), "event"); ck(hip.hipEventCreate(ctypes.byref(ev1)), "event")
    HIP_LAUNCH_PARAM_BUFFER_POINTER, HIP_LAUNCH_PARAM_BUFFER_SIZE, HIP_LAUNCH_PARAM_END = 1, 2, 3

    def launch(fn, waves_per_simd, reps):
        arg = (ctypes.c_uint32 * 2)(reps, 0)
        size = ctypes.c_size_t(8)
        extra = (ctypes.c_void_p * 5)(HIP_LAUNCH_PARAM_BUFFER_POINTER, ctypes.cast(arg, ctypes.c_void_p), HIP_LAUNCH_PARAM_BUFFER_SIZE,
                                      ctypes.cast(ctypes.pointer(size), ctypes.c_void_p), HIP_LAUNCH_PARAM_END)
        # 256 CUs x 4 SIMDs: workgroups of 256 threads (one wavefront on each SIMD of a CU), waves_per_simd workgroups per CU
        ck(hip.hipEventRecord(ev0, None), "record")
        ck(hip.hipModuleLaunchKernel(fn, 256 * waves_per_simd, 1, 1, 256, 1, 1, 0, None, None, extra), "launch")
        ck(hip.hipEventRecord(ev1, None), "record")
        ck(hip.hipEventSynchronize(ev1), "sync")
        ms = ctypes.c_float()
        ck(hip.hipEventElapsedTime(ctypes.byref(ms), ev0, ev1), "elapsed")
        return ms.value

    fns = {}
    for n in (t[0] for t in INSTR):
        f = ctypes.c_void_p()
        ck(hip.hipModuleGetFunction(ctypes.byref(f), mod, ("ic_" + n).encode()), n)
        fns[n] = f
    res = {}
    for w in (1, 2, 4):
        for n, body in ((t[0], t[1]) for t in INSTR):
            launch(fns[n], w, 200)
            t = min(launch(fns[n], w, reps) for _ in range(3))
            res[(n, w)] = t / (body.count("\n") + 1)
    print("# cycles per instruction and wavefront slot, relative to v_fma_f64 = 4 at the same occupancy; ms of %d x %d copies in brackets" % (reps, COPIES))
    print("%-22s %16s %16s %16s" % ("instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD"))
    for n in (t[0] for t in INSTR):
        cells = []
        for w in (1, 2, 4):
            ref = res[("v_fma_f64", w)]
            cells.append("%6.2f (%7.3f)" % (4.0 * res[(n, w)] / ref, res[(n, w)]))
        print("%-22s %16s %16s %16s" % (n, *cells))
    # absolute: the fma loop gives the clock
    for w in (1, 2, 4):
        t = res[("v_fma_f64", w)]
        print("# v_fma_f64 at %d wave(s)/SIMD: %.3f ms for %d instructions per wavefront -> %.2f GHz if 4 cycles each and wavefronts of a SIMD alternate"
              % (w, t, reps * COPIES, reps * COPIES * w * 4 / (t * 1e-3) / 1e9))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit")
    ap.add_argument("--run")
    ap.add_argument("--reps", type=int, default=20000)
    a = ap.parse_args()
    if a.emit:
        emit(a.emit)
    if a.run:
        run(a.run, a.reps)
