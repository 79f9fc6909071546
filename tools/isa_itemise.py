#!/usr/bin/env python3
"""Itemised static instruction mix of the innermost loops (the MH iteration bodies: one basic-block loop per entry-count shape) of one
kernel in a hipcc -S listing — what the vector-ALU slots of an iteration are spent on BESIDES fp64 arithmetic.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o sl.s pyhillfit_amd/csrc/phf_single_level.hip
    python tools/isa_itemise.py sl.s mh_advance_kernelILi2ELb0ELi2E            (produced profiles/r05/c3_valu_itemised.txt)

A loop = a backward branch whose span holds no other backward branch; classes by opcode (and, for Philox, by its constants)."""
import collections
import re
import sys

CLASSES = [
    ("fp64 arithmetic (fma, mul, add: what the PMC's flop counters see)", r"v_(fma|fmac|mul|add)_f64"),
    ("fp64 estimates (v_rcp_f64, v_rsq_f64: quarter rate)", r"v_(rcp|rsq|sqrt)_f64"),
    ("fp64 other: min / max (clamps)", r"v_(min|max)_f64"),
    ("fp64 other: compares", r"v_cmp\w*_f64"),
    ("fp64 other: ldexp, conversions", r"v_(ldexp_f64|cvt_f64_\w+|cvt_\w+_f64|frexp\w*_f64|trunc_f64|rndne_f64)"),
    ("Philox: 32 x 32 -> 64 multiplies", r"v_(mad_u64_u32|mul_hi_u32|mul_lo_u32)"),
    ("Philox / bit logic: 3-input xor, xor, and-or, bfi", r"v_(bitop3_b32|xor_b32|and_or_b32|bfi_b32|or3_b32|xad_u32)"),
    ("selects (v_cndmask_b32: two per double)", r"v_cndmask_b32"),
    ("moves (v_mov_b32 / _b64, v_accvgpr, lane ops)", r"v_(mov_b32|mov_b64|accvgpr_\w+|readlane\w*|writelane\w*|readfirstlane\w*|pk_mov_b32)"),
    ("integer / address arithmetic (table indices, exponent fields)", r"v_(and_b32|or_b32|not_b32|lshl\w*|lshr\w*|ashr\w*|add\w*_u32|add_co\w*|addc\w*|sub\w*_u32|subrev\w*|mul_u32_u24|mad_u32_u24|min_u32|max_u32|min_i32|max_i32|lshl_add_u64|bfe_\w+|cvt_\w*32\w*|cmp\w*_[ui]32|cmp\w*_[ui]64|alignbit\w*|perm_b32|add3_u32|lshl_or_b32|med3\w*)"),
]


# Issue cost in cycles per wavefront instruction with two or more wavefronts per SIMD, measured (tools/isa_instr_cost.py,
# profiles/r05/gfx950_instruction_costs.txt): quarter-rate estimates 16; the simple 32-bit VOP1 / VOP2 operations 2 — when encoded _e32 and
# without a scalar-register operand (with one they measure 4) —; everything else, every VOP3 encoding included, 4.
FAST_E32 = r"v_(mov_b32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|lshrrev_b32|ashrrev_i32|accvgpr_read_b32|accvgpr_write_b32|fma_f32|add_f32|mul_f32)_e32"


def issue_cycles(line):
    op = line.split()[0]
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return 16
    if op.startswith("v_accvgpr") or re.fullmatch(FAST_E32, op):
        operands = line.split(None, 1)[1] if " " in line else ""
        if not re.search(r"\bs\d+|\bs\[|\bvcc|\bexec|\bm0", operands):
            return 2
    return 4


# pairs of the Crumb set per (uncensored entries, censored entries) after merging replicates (pyhillfit_amd.doseresponse.pack_single_level)
SHAPE_WEIGHTS = [((4, 1), 45), ((4, 0), 41), ((4, 2), 25), ((4, 4), 22), ((4, 3), 13), ((3, 4), 11), ((2, 4), 9), ((3, 3), 7), ((1, 4), 6),
                 ((2, 3), 5), ((2, 2), 5), ((3, 2), 4), ((3, 1), 3), ((1, 3), 3), ((2, 1), 3), ((2, 0), 1)]


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    starts = [i for i, l in enumerate(src) if re.match(r"_Z\w+:", l)]
    for si, s0 in enumerate(starts):
        name = src[s0].split(":", 1)[0]
        if want not in name:
            continue
        end = starts[si + 1] if si + 1 < len(starts) else len(src)
        labels, ins = {}, []
        for l in src[s0 + 1:end]:
            l = l.strip()
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = len(ins)
                continue
            if l.startswith(".Lfunc_end"):
                break
            if not l or l.startswith(";") or l.startswith("."):
                continue
            ins.append(l)
        back = []
        for i, l in enumerate(ins):
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
            if m:
                lab = m.group(1) or m.group(2)
                if lab in labels and labels[lab] <= i:
                    back.append((labels[lab], i))
        inner = [(a, b) for a, b in back if not any(a <= c and d <= b and (c, d) != (a, b) for c, d in back)]
        inner = [(a, b) for a, b in inner if b - a > 300]                  # the iteration bodies (the point loops of the generic body are short)
        print("%s: %d instructions, %d iteration bodies" % (name, len(ins), len(inner)))
        rows = []
        for a, b in inner:
            body = ins[a:b + 1]
            c = collections.Counter()
            for l in body:
                op = l.split()[0]
                if not op.startswith("v_"):
                    c["(not VALU) " + ("s_nop" if op == "s_nop" else "scalar" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
                                       "scratch" if op.startswith("scratch_") else "global memory" if op.startswith("global_") or op.startswith("buffer_") else op)] += 1
                    continue
                for cname, pat in CLASSES:
                    if re.fullmatch(pat + r"(_e32|_e64|_dpp|_sdwa)?", op):
                        c[cname] += 1
                        break
                else:
                    c["other VALU: " + op] += 1
            valu = sum(v for k_, v in c.items() if not k_.startswith("(not VALU)"))
            fp = sum(1 for l in body if re.match(r"v_(fma|fmac|mul|add)_f64", l.split()[0]))
            cyc = collections.Counter()
            for l in body:
                if l.startswith("v_"):
                    cyc["fp64 fma / mul / add" if re.match(r"v_(fma|fmac|mul|add)_f64", l) else "quarter-rate estimates" if issue_cycles(l) == 16 else
                        "2-cycle 32-bit operations" if issue_cycles(l) == 2 else "other 4-cycle instructions"] += issue_cycles(l)
            c["(issue cycles)"] = cyc
            rows.append((valu, fp, a, b, c))
        # which entry-count shape a body is (single-level kernels): LDS reads = 13 + uncensored entries + 6 x censored entries
        # (one 2^(j/64) read per entry; five 16-byte reads of the log Phi table per censored entry; 13 for the draws and the logarithms)
        weights = dict(SHAPE_WEIGHTS) if len(sys.argv) > 3 and sys.argv[3] == "crumb" else {}
        tot, wsum = collections.Counter(), 0
        cyc_tot = collections.Counter()
        for valu, fp, a, b, c in sorted(rows, key=lambda r: r[:4]):
            cyc = c.pop("(issue cycles)")
            lds = c.get("(not VALU) LDS", 0)
            shape = next(((ko, kc) for ko in range(1, 5) for kc in range(0, 5) if 13 + ko + 6 * kc == lds), None)
            w = weights.get(shape, 0)
            print("\n  body [%d..%d]%s: %d instructions, %d VALU of which %d fp64 fma / mul / add" % (
                a, b, "" if shape is None else " = %d uncensored + %d censored entries (%d of the 210 Crumb pairs)" % (shape[0], shape[1], w), b - a + 1, valu, fp))
            for k_, v in sorted(c.items(), key=lambda kv: (kv[0].startswith("(not VALU)"), -kv[1])):
                print("     %5d  %s" % (v, k_))
                tot[k_] += v * w
            print("     issue cycles of the vector pipe per wavefront: %d = %s" % (sum(cyc.values()), ", ".join("%d %s" % (v, k_) for k_, v in cyc.most_common())))
            for k_, v in cyc.items():
                cyc_tot[k_] += v * w
            wsum += w
        if wsum:
            print("\n  AVERAGE over the %d Crumb pairs whose shape has a straight-line body (weights = pairs per shape):" % wsum)
            valu = sum(v for k_, v in tot.items() if not k_.startswith("(not VALU)")) / wsum
            print("     %7.1f  VALU in all" % valu)
            for k_, v in sorted(tot.items(), key=lambda kv: (kv[0].startswith("(not VALU)"), -kv[1])):
                print("     %7.1f  %s" % (v / wsum, k_))
            print("     issue cycles of the vector pipe per wavefront and iteration (measured cost per instruction: tools/isa_instr_cost.py): %.0f" % (sum(cyc_tot.values()) / wsum))
            for k_, v in cyc_tot.most_common():
                print("     %7.0f  %s" % (v / wsum, k_))


if __name__ == "__main__":
    main()
