#!/usr/bin/env python
"""Where the wall time of ONE keyframe goes when nothing overlaps it (VERDICT r4 next #1a): from a rocprofv3 kernel trace (csv) of

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 40 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api

take the steady-state keyframes (one cost-volume sad launch each), and per launch position within the keyframe report the kernel, its
median duration and the median GAP in front of it (start - end of the previous launch on the device: dispatch latency of a dependent launch,
plus whatever the host failed to enqueue in time); sums per kernel family and for the whole keyframe.  Writes a JSON summary.

    python tools/trace_gaps.py DIR/.../t_kernel_trace.csv [--out profiles/r05_c2_launch_gaps.json] [--plan-names names.json]
"""
import argparse
import csv
import json
import statistics
import sys


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return k[:i].strip()
    return k.strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-last", type=int, default=12, help="keyframes at the end of the trace to ignore (bench.py finishes with per-layer timing passes)")
    ap.add_argument("--keyframes", type=int, default=20)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "cv_sad" in r[2]]
    if len(marks) < a.skip_last + a.keyframes + 2:
        print("too few keyframes in the trace", len(marks))
        return 1
    # a keyframe = the launches from one cost-volume statistics prepass (or sad launch) to the next
    starts = [i for i, r in enumerate(rows) if "cv_kf_stats" in r[2]] or marks
    sel = starts[-(a.skip_last + a.keyframes + 1):-a.skip_last]
    frames = [rows[b:e] for b, e in zip(sel, sel[1:])]
    n = statistics.mode(len(f) for f in frames)
    frames = [f for f in frames if len(f) == n and [x[2] for x in f] == [x[2] for x in frames[0]]] or [f for f in frames if len(f) == n]
    # the launch order of --single-stream: cv stage first, then encoder, then main (model._submit_locked); rotate nothing - report as traced
    per = []
    for pos in range(n):
        dur = statistics.median((f[pos][1] - f[pos][0]) / 1e3 for f in frames)
        gap = statistics.median(((f[pos][0] - f[pos - 1][1]) / 1e3) for f in frames) if pos else None
        per.append({"pos": pos, "kernel": frames[0][pos][2], "us": round(dur, 2), "gap_before_us": None if gap is None else round(gap, 2)})
    period = statistics.median((b[0][0] - a_[0][0]) / 1e3 for a_, b in zip(frames, frames[1:])) if len(frames) > 1 else None
    ksum = sum(p["us"] for p in per)
    gsum = sum(p["gap_before_us"] or 0.0 for p in per)
    fam = {}
    for p in per:
        key = p["kernel"].split("<")[0]
        d = fam.setdefault(key, {"launches": 0, "us": 0.0, "gap_before_us": 0.0})
        d["launches"] += 1
        d["us"] += p["us"]
        d["gap_before_us"] += p["gap_before_us"] or 0.0
    for d in fam.values():
        d["us"], d["gap_before_us"] = round(d["us"], 1), round(d["gap_before_us"], 1)
    big = sorted((p for p in per if (p["gap_before_us"] or 0) > 4.0), key=lambda p: -p["gap_before_us"])[:12]
    out = {"keyframes_used": len(frames), "launches_per_keyframe": n, "kernel_us_per_keyframe": round(ksum, 1), "gap_us_per_keyframe": round(gsum, 1),
           "keyframe_period_us": None if period is None else round(period, 1),
           "between_keyframes_us": None if period is None else round(period - ksum - gsum, 1),
           "median_gap_us": round(statistics.median(p["gap_before_us"] for p in per[1:]), 2),
           "largest_gaps": big, "by_kernel_family": dict(sorted(fam.items(), key=lambda kv: -kv[1]["us"])), "launches": per,
           "note": "one keyframe at a time, all stages on one stream (bench.py --in-flight 1 --single-stream) under rocprofv3 --kernel-trace: "
                   "gap = start of a launch - end of the launch in front of it; between_keyframes = period - kernels - gaps (host turnaround)"}
    print(json.dumps({k: v for k, v in out.items() if k not in ("launches", "by_kernel_family")}, indent=1))
    for k, v in out["by_kernel_family"].items():
        print(f"  {k:34s} x{v['launches']:3d}  {v['us']:8.1f} us  gaps in front {v['gap_before_us']:7.1f} us")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
