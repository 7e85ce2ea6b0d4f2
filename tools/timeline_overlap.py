#!/usr/bin/env python
"""Offline look at a rocprofv3 kernel trace (csv written by tools/sessions/r03_s4.sh: queue, stream, kernel, start, end in ns):
how busy the device is, how many kernels run at once, per-stream idle gaps, and the keyframe period.

    python tools/timeline_overlap.py gpurun_out/r03_s4/t1.csv
"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = [(r["queue"], r["stream"], r["kernel"], int(r["start"]), int(r["end"])) for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: r[3])
    # steady state of the timed loop: 10 keyframes well before the end (bench.py finishes with time_layers: 5 passes of one launch at a
    # time on the current stream, which must not be mistaken for the pipeline)
    cv = [r for r in rows if "cv_sad_march" in r[2]]
    if len(cv) < 40:
        print("too few keyframes"); return
    t0, t1 = cv[-30][3], cv[-20][3]
    period = (t1 - t0) / 10 / 1e3
    win = [r for r in rows if r[3] >= t0 and r[3] < t1]
    # concurrency sweep
    ev = []
    for r in win:
        ev.append((r[3], 1)); ev.append((min(r[4], t1), -1))
    ev.sort()
    busy = defaultdict(int)
    cur, last = 0, t0
    for t, d in ev:
        busy[cur] += t - last
        cur += d; last = t
    busy[cur] += t1 - last
    tot = t1 - t0
    ksum = sum(min(r[4], t1) - r[3] for r in win)
    print(f"{path}: period {period:.1f} us/keyframe = {1e6/period:.0f} keyframes/s; sum of kernel durations {ksum/10/1e3:.0f} us/keyframe")
    print("  time with n kernels running:", {k: f"{100*v/tot:.1f}%" for k, v in sorted(busy.items())})
    streams = defaultdict(list)
    for r in win:
        streams[(r[0], r[1])].append(r)
    for key, rs in sorted(streams.items(), key=lambda kv: -len(kv[1])):
        dur = sum(r[4] - r[3] for r in rs)
        gaps = [b[3] - a[4] for a, b in zip(rs, rs[1:])]
        small = [g for g in gaps if g < 20000]
        print(f"  queue {key[0]} stream {key[1]}: {len(rs)/10:.1f} kernels/keyframe, busy {dur/10/1e3:.0f} us/keyframe, "
              f"median gap {sorted(gaps)[len(gaps)//2]/1e3 if gaps else 0:.1f} us, gaps<20us sum {sum(small)/10/1e3:.0f} us/keyframe, "
              f"big gaps sum {sum(g for g in gaps if g >= 20000)/10/1e3:.0f} us/keyframe")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
