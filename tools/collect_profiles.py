#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last GPU pass from gpurun_out/ (scratch) into profiles/<round>/ (tracked) and
refresh profiles/pmc_facts.json from the PMC summary.

  python tools/collect_profiles.py --round r04 --tag v40 --workload c5_moments --chains 6881280 --iters 500 [--kernel mh_advance_kernel<2,]
(--workload is the NAME of a bench region = the key of pmc_facts.json: c2 | c3 | c4 | c5 | c5_moments | c3_model1; tools/gpu_r04_prof.sh)

Facts per workload: fp64 flop per MH iteration = (2 FMA + MUL + ADD + TRANS) / (waves x iterations) and HBM traffic per
launch = WRITE_SIZE + 2 x FETCH_SIZE (KiB counters; the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)."""
import argparse
import glob
import json
import os
import re
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_summary(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=(\d+) mean=(\S+)", line)
            if m and cur is not None:
                out[cur][m.group(1)] = float(m.group(3))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--chains", type=int, required=True, help="chains of the launch (all problems)")
    ap.add_argument("--iters", type=int, required=True, help="MH iterations per launch")
    ap.add_argument("--thinning", type=int, default=5)
    ap.add_argument("--kernel", default=None, help="prefix of the kernel's name in the PMC summary (default: the one with most VALU instructions)")
    ap.add_argument("--note", default="")
    ap.add_argument("--bench-args", default="", help="the bench.py arguments of the region (recorded in the facts' source line)")
    a = ap.parse_args()
    dst = os.path.join(REPO, "profiles", a.round)
    os.makedirs(dst, exist_ok=True)
    w = a.workload
    # rocprofv3 --kernel-trace --stats of the same bench command: newest run directory
    stats = sorted(glob.glob(os.path.join(REPO, "gpurun_out", "prof_%s" % w, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, "%s_rocprof_kernel_stats_%s.csv" % (w, a.tag)))
    summ = os.path.join(REPO, "gpurun_out", "pmc_%s_summary.txt" % w)
    if not os.path.exists(summ):
        print("no PMC summary for", w); return
    pmc_name = "%s_pmc_%s.txt" % (w, a.tag)
    shutil.copy(summ, os.path.join(dst, pmc_name))
    ks = parse_summary(summ)
    kernels = [k for k in ks if "SQ_INSTS_VALU" in ks[k]]
    key = max(kernels, key=lambda k: ks[k]["SQ_INSTS_VALU"]) if a.kernel is None else [k for k in kernels if k.startswith(a.kernel)][0]
    # a workload made of several kernels (c4: one per Ne group): sums over the kernels
    group = [k for k in kernels if "advance" in k] if w.startswith("c4") else [key]
    tot = lambda c: sum(ks[k].get(c, 0.0) for k in group)
    # 64 chains = one wavefront-load of work per launch, whether a wavefront of its own runs it (SQ_WAVES = chains / 64) or the
    # persistent wavefronts of a queued launch do (SQ_WAVES = the grid)
    waves = a.chains / 64.0
    per = lambda c: tot(c) / (waves * a.iters)
    fma, mul, add, tr = per("SQ_INSTS_VALU_FMA_F64"), per("SQ_INSTS_VALU_MUL_F64"), per("SQ_INSTS_VALU_ADD_F64"), per("SQ_INSTS_VALU_TRANS_F64")
    facts_path = os.path.join(REPO, "profiles", "pmc_facts.json")
    facts = json.load(open(facts_path)) if os.path.exists(facts_path) else {}
    facts[w] = {
        "source": "profiles/%s/%s (rocprofv3 --pmc on `bench.py %s`, kernel(s) %s)%s" % (a.round, pmc_name, a.bench_args or ("--workload " + w), ", ".join(g[:40] for g in group), (" " + a.note) if a.note else ""),
        "chains": a.chains, "iterations_per_launch": a.iters, "thinning": a.thinning,
        "flop_per_iteration": round(2 * fma + mul + add + tr, 1),
        "traffic_bytes_per_launch": int(round((tot("WRITE_SIZE") + 2 * tot("FETCH_SIZE")) * 1024)),
        "derivation": "flop = (2*SQ_INSTS_VALU_FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64) / (%d blocks of 64 chains * %d iterations) = (2*%.1f+%.1f+%.1f+%.1f); "
                      "traffic = WRITE_SIZE %.1f KiB + 2 x FETCH_SIZE %.1f KiB" % (waves, a.iters, fma, mul, add, tr, tot("WRITE_SIZE"), tot("FETCH_SIZE")),
        "instruction_mix_per_iteration": {"VALU": round(per("SQ_INSTS_VALU"), 1), "of which fp64 arithmetic": round(fma + mul + add + tr, 1),
                                          "SALU": round(per("SQ_INSTS_SALU"), 1), "SMEM": round(per("SQ_INSTS_SMEM"), 1), "LDS": round(per("SQ_INSTS_LDS"), 1),
                                          "wave_cycles": round(4 * per("SQ_WAVE_CYCLES"), 0),
                                          "wait_inst_any_frac": round(tot("SQ_WAIT_INST_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), 3),
                                          "wait_any_frac": round(tot("SQ_WAIT_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), 3),
                                          "valu_active_frac_of_wave_cycles": round(tot("SQ_ACTIVE_INST_VALU") / max(tot("SQ_WAVE_CYCLES"), 1), 3),
                                          "lds_active_quad_cycles": round(per("SQ_ACTIVE_INST_LDS"), 1),
                                          "lds_bank_conflict_cycles": round(per("SQ_LDS_BANK_CONFLICT"), 1),
                                          "lds_idx_active_cycles": round(per("SQ_LDS_IDX_ACTIVE"), 1)},
    }
    json.dump(facts, open(facts_path, "w"), indent=1)
    print(json.dumps(facts[w], indent=1))


if __name__ == "__main__":
    main()
