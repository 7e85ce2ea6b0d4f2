#!/usr/bin/env python
"""Condense rocprofv3 result databases (gpurun_out/...) into the small tracked summaries under profiles/.

    python tools/summarize_prof.py --tag r01 --stats gpurun_out/prof/r01_results.db \
        --pmc gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db ...
"""
import argparse
import collections
import csv
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_kernel_name(k):
    """'void (anonymous namespace)::cv_sad_kernel<32, 16, 1, 0>((anonymous namespace)::CvArgs)' -> 'cv_sad_kernel<32, 16, 1, 0>':
    drop the namespace and return type first, THEN cut the parameter list (round 1 cut at the first '(' and collapsed every
    anonymous-namespace kernel into one nameless bucket)."""
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(k):           # the parameter list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return k[:i].strip()
    return k.strip()


def family_metrics(agg, match):
    """Counters summed over every kernel whose name contains `match`, with the derived figures the bench line quotes."""
    tot = collections.defaultdict(float)
    n = 0
    for k, v in agg.items():
        if match in k:
            for cn, d in v.items():
                tot[cn] += d["sum"]
            n = max(n, max(d["dispatches"] for d in v.values()))
    if not n:
        return None
    d = {"dispatches": n}
    for cn in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES",
               "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE",
               "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if cn in tot:
            d[cn + "_per_dispatch"] = tot[cn] / n
    if "SQ_LDS_BANK_CONFLICT" in tot and tot.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        d["hbm_bytes_per_dispatch_raw"] = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / n
    if "SQ_ACTIVE_INST_VALU" in tot and "SQ_BUSY_CYCLES" in tot and tot["SQ_BUSY_CYCLES"]:
        d["note_valu"] = "SQ_INSTS_VALU = wave-level VALU instructions; x64 lanes / 78.6e12 lane-instr/s = VALU-issue floor of the launch"
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats", default=None)
    ap.add_argument("--stats-seq", default=None, help="kernel trace of the same command with --in-flight 1 (no overlap between keyframes): "
                                                      "kernel-only durations that are not inflated by a second keyframe's kernels")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--bench-line", default=None, help="JSON line of bench.py from the profiled session: its roofline.profile_stamp.running_plan becomes the stamp of this set")
    ap.add_argument("--launches-per-step", type=int, default=None, help="dispatches of each kernel instance per profiled run / steps")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    if args.stats:
        c = sqlite3.connect(args.stats)
        cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
        rows = list(c.execute("select * from top_kernels"))
        with open(os.path.join(out_dir, f"{args.tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols + ["note: durations in us; rocprofv3 --kernel-trace --stats"])
            w.writerows(rows)
    if args.stats_seq:
        c = sqlite3.connect(args.stats_seq)
        cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
        rows = list(c.execute("select * from top_kernels"))
        with open(os.path.join(out_dir, f"{args.tag}_kernel_stats_seq.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols + ["note: durations in us; rocprofv3 --kernel-trace --stats; bench.py --in-flight 1 (one keyframe at a time)"])
            w.writerows(rows)
    # every profile set carries the stamp of the plan it was taken on (Plan.launch_stamp() of the profiled run, read from the bench
    # line of the same session): bench.py refuses to quote a stale one
    import sys
    sys.path.insert(0, ROOT)
    from monorec_amd import _lib
    stamp = None
    if args.bench_line and os.path.exists(args.bench_line):
        try:
            line = json.loads(open(args.bench_line).read().strip().splitlines()[-1])
            stamp = line["roofline"]["profile_stamp"]["running_plan"]
        except Exception:
            stamp = None
    with open(os.path.join(out_dir, f"{args.tag}_stamp.json"), "w") as f:
        json.dump({"plan_stamp": stamp, "abi": _lib.MR_ABI_VERSION,
                   "note": "Plan.launch_stamp() of the profiled run = sha256(kernel family, variant and schedule of every convolution launch + ABI)[:16], "
                           "from the bench line of the session that took the traces / counters of this tag"}, f)
    agg = collections.defaultdict(dict)
    for path in args.pmc:
        c = sqlite3.connect(path)
        q = "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"
        for k, cn, n, s in c.execute(q):
            agg[k][cn] = {"dispatches": n, "sum": s, "per_dispatch": s / n}
    if agg:
        summary = {}
        conv = collections.defaultdict(float)
        conv_n = 0
        for k, v in agg.items():
            short = short_kernel_name(k)
            summary[short] = v
            if "conv_mfma_kernel" in k or "conv_b8_kernel" in k or "conv3x3_wino" in k or "convt4x4_wino" in k or "conv1d3_wino" in k or "conv1d_ct_kernel" in k or "upconv2x2_wino" in k:
                for cn, d in v.items():
                    conv[cn] += d["sum"]
                conv_n += next(iter(v.values()))["dispatches"]
        derived = {"conv_mfma_kernel_all_instances": {"dispatches": conv_n}}
        d = derived["conv_mfma_kernel_all_instances"]
        if "FETCH_SIZE" in conv:
            d["FETCH_SIZE_KiB_per_dispatch"] = conv["FETCH_SIZE"] / conv_n
        if "WRITE_SIZE" in conv:
            d["WRITE_SIZE_KiB_per_dispatch"] = conv["WRITE_SIZE"] / conv_n
        if "FETCH_SIZE" in conv and "WRITE_SIZE" in conv:
            d["hbm_bytes_per_dispatch_raw"] = (conv["FETCH_SIZE"] + conv["WRITE_SIZE"]) * 1024 / conv_n
            d["hbm_bytes_per_dispatch"] = (2 * conv["FETCH_SIZE"] + conv["WRITE_SIZE"]) * 1024 / conv_n
            d["note"] = ("inputs and weights are staged with 16 B/lane LDS-DMA: on gfx950 rocprofv3 FETCH_SIZE reports half the "
                         "bytes of such streams (MI355X_MICROARCH.md, HBM section), so hbm_bytes_per_dispatch = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024; _raw is uncorrected; WRITE_SIZE as reported")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in conv and "GRBM_GUI_ACTIVE" in conv:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            d["mfma_util"] = (conv["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (conv["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_WAIT_ANY" in conv and "SQ_WAVE_CYCLES" in conv:
            d["wait_any_frac_of_wave_cycles"] = conv["SQ_WAIT_ANY"] / conv["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in conv:
            d["lds_bank_conflict_cycles"] = conv["SQ_LDS_BANK_CONFLICT"]
        for fam in ("cv_sad", "cv_fuse", "depth_heads", "mask_classifier"):
            m = family_metrics(agg, fam)
            if m:
                derived[fam + "_kernels"] = m
        with open(os.path.join(out_dir, f"{args.tag}_pmc_summary.json"), "w") as f:
            json.dump({"derived": derived, "per_kernel": summary}, f, indent=1, sort_keys=True)
        print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
