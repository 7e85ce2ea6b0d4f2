#!/usr/bin/env python
"""Every measured table for ONE input shape in one go (run on the GPU box; VERDICT r5 #6: round 5 did this by hand in tools/sessions/r05_s18.sh):

    python tools/tune_all.py --shape B H W F D [--bf16] [--out-dir gpurun_out/tables] [--install]

Runs, on COPIES of the committed tables, the five tuners in the order their results depend on each other - direct-kernel schedules (tools/tune_conv.py,
LDS ring depths included), 3x3 forms (tools/bench_wino.py), ConvTranspose2d(4,2) forms (tools/bench_wino_t.py), 1-D forms (tools/bench_wino1d.py),
stride-2 pairs (tools/bench_stride2.py) - or, with --bf16, the B8 schedule tuner (tools/tune_b8.py).  Only signatures without an entry are measured
(--all re-measures every signature of the shape).  --install copies the results over monorec_amd/tuned_*.json.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=5, metavar=("B", "H", "W", "F", "D"), required=True)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--all", action="store_true", help="re-measure signatures that already have an entry")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "tables"))
    ap.add_argument("--install", action="store_true")
    ap.add_argument("--timeout", type=int, default=900)
    a = ap.parse_args()
    b, h, w, f, d = a.shape
    os.makedirs(a.out_dir, exist_ok=True)
    names = ("tuned_schedules.json", "tuned_winograd.json", "tuned_b8.json")
    for n in names:
        dst = os.path.join(a.out_dir, n)
        if not os.path.exists(dst):
            shutil.copy(os.path.join(ROOT, "monorec_amd", n), dst)
    sched, wino, b8 = (os.path.join(a.out_dir, n) for n in names)
    shape = ["--batch", str(b), "--height", str(h), "--width", str(w), "--frames", str(f), "--depths", str(d)]
    env = dict(os.environ, MR_TUNED_SCHEDULES=sched, MR_TUNED_WINOGRAD=wino, MR_TUNED_B8=b8)
    py = sys.executable
    if a.bf16:
        steps = [("tune_b8", [py, "tools/tune_b8.py"] + shape + ["--emit", b8])]
    else:
        steps = [("tune_conv", [py, "tools/tune_conv.py"] + shape + ["--merge", "--out", sched] + ([] if a.all else ["--missing"])),
                 ("bench_wino", [py, "tools/bench_wino.py"] + shape + ["--emit", wino]),
                 ("bench_wino_t", [py, "tools/bench_wino_t.py"] + shape + ["--emit", wino]),
                 ("bench_wino1d", [py, "tools/bench_wino1d.py"] + shape + ["--emit", wino]),
                 ("bench_stride2", [py, "tools/bench_stride2.py"] + shape + ["--emit", wino])]
    before = {n: len(json.load(open(os.path.join(a.out_dir, n)))) for n in names}
    for tag, cmd in steps:
        t0 = time.time()
        log = os.path.join(a.out_dir, f"{tag}_b{b}_{h}x{w}_f{f}_d{d}.log")
        with open(log, "w") as lf:
            rc = subprocess.run(cmd, cwd=ROOT, env=env if tag != "tune_conv" else dict(env, MR_TUNED_SCHEDULES=""), stdout=lf, stderr=subprocess.STDOUT, timeout=a.timeout).returncode
        tail = open(log).read().strip().splitlines()[-1:] or [""]
        print(f"{tag}: rc={rc} {time.time() - t0:.0f} s  {tail[0][:200]}", flush=True)
    after = {n: len(json.load(open(os.path.join(a.out_dir, n)))) for n in names}
    print("entries:", {n: (before[n], after[n]) for n in names})
    if a.install:
        for n in names:
            shutil.copy(os.path.join(a.out_dir, n), os.path.join(ROOT, "monorec_amd", n))
        print("installed into monorec_amd/")


if __name__ == "__main__":
    main()
