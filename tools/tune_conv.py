#!/usr/bin/env python
"""Measure every launchable (MB, NB, split_k, CK) schedule of every convolution of a MonoRec launch plan on
the attached MI355X and write the fastest per layer signature to monorec_amd/tuned_schedules.json
(consulted by engine.choose_schedule).  Run on the GPU box:

    python tools/tune_conv.py [--batch 1 --height 256 --width 512 --frames 2 --depths 32] [--merge]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from monorec_amd import engine, synth  # noqa: E402
from monorec_amd.model import MonoRecModel  # noqa: E402

DEV = "cuda:0"


def time_op(fn, reps=5, warm=2):
    stream = torch.cuda.current_stream()
    for _ in range(warm):
        fn(stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn(stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_op_streams(fn, streams, reps=5, warm=2):
    """Per-launch time with the same launch repeated round-robin on several streams: what a layer costs when
    keyframes are in flight side by side (head / tail bubbles of one launch filled by the other)."""
    base = torch.cuda.current_stream()
    for s in streams:
        for _ in range(warm):
            fn(s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(base)
    for s in streams:
        s.wait_event(e0)
    for _ in range(reps):
        for s in streams:
            fn(s.cuda_stream)
    for s in streams:
        e = torch.cuda.Event()
        e.record(s)
        base.wait_event(e)
    e1.record(base)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * len(streams))


def build_candidate(spec, sched, tensors, bf16=False):
    plan = engine.Plan.bare(DEV, schedule_override={"t": sched}, bf16=bf16)
    srcs, out, res, weight, bias, phase_w = tensors
    phases = None
    if spec["phases"] is not None:
        phases = [(phase_w[i], *spec["phases"][i][:4]) for i in range(len(spec["phases"]))]
    plan.conv("main", "t", srcs, weight, bias, out, stride=spec["stride"], pad=spec["pad"], grid=spec["grid"],
              act=spec["act"], p0=spec["p0"], p1=spec["p1"], in_mode=spec["in_mode"], tf=spec["tf"],
              residual=res, out_step=spec["out_step"], out_off=spec["out_off"], phases=phases)
    plan.finalize()
    return plan, plan.stages["main"][0][1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--merge", action="store_true", help="keep entries already in the table for other shapes")
    ap.add_argument("--max-cands", type=int, default=2000)
    ap.add_argument("--streams", type=int, default=1, help="time each candidate on this many concurrent streams")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lds-cap", type=int, default=160 * 1024)
    ap.add_argument("--finalists", type=int, default=4, help="candidates re-timed in the second pass")
    ap.add_argument("--final-reps", type=int, default=20, help="launches per round of the second pass (0 = skip it)")
    ap.add_argument("--bf16", action="store_true", help="tune the MR_COMPUTE_BF16 launches (hip_bf16=True plans)")
    ap.add_argument("--bf16x3", action="store_true", help="tune the MR_COMPUTE_BF16X3 launches (hip_bf16x3=True plans)")
    ap.add_argument("--only", default=None, help="tune only the layers whose name contains one of these comma-separated strings")
    ap.add_argument("--missing", action="store_true", help="tune only the layer signatures the table has no entry for")
    ap.add_argument("--prefer-nosplit", type=float, default=None,
                    help="take the fastest schedule WITHOUT split_k (K split across the waves included) when it is within this many per cent of "
                         "the overall fastest: one launch less per layer (no finishing kernel, no workspace round trip)")
    ap.add_argument("--out", default=os.path.join(ROOT, "monorec_amd", "tuned_schedules.json"))
    ap.add_argument("--report", default=None)
    args = ap.parse_args()

    model = MonoRecModel(cv_depth_steps=args.depths)
    sd = synth.seeded_state_dict(model.state_dict())
    engine.TUNED.clear()
    plan = engine.Plan(sd, args.batch, args.height, args.width, args.frames, args.depths, (0.33, 0.0025), "cpu", bf16=2 if args.bf16x3 else int(args.bf16), winograd=False)
    table = {}
    if args.merge and os.path.exists(args.out):
        table = json.load(open(args.out))
    report = []
    t_start = time.time()
    seen = set()
    streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None
    g = torch.Generator().manual_seed(0)
    for c in plan.conv_log:
        if c["sig"] is None or c["sig"] in seen:
            continue
        if args.only and not any(tok in c["name"] for tok in args.only.split(",")):
            continue
        if args.missing and c["sig"] in table:
            continue
        seen.add(c["sig"])
        spec = c["spec"]
        cout, cin, kh, kw = spec["w_shape"]
        src_channels = [s[1] for s in spec["src_shapes"]]
        nph = 1 if spec["phases"] is None else len(spec["phases"])
        cands = engine.candidate_schedules(cout, src_channels, kh, kw, spec["stride"][0], spec["stride"][1],
                                           spec["grid"][0], spec["grid"][1], c["batch"], nph, lds_cap=args.lds_cap, bf16=c.get("bf16", False))
        # prune: split-K only while the launch is short of ~8 workgroups per CU; drop tiny grids
        keep = []
        for cd in cands:
            base = cd["wgs"] // cd["split_k"]
            if cd["split_k"] > 1 and base >= 1536:
                continue
            if cd["wgs"] > 16384:
                continue
            keep.append(cd)
        keep.sort(key=lambda cd: (-min(cd["wgs"], 1024), -cd["mb"] * cd["nb"]))
        keep = keep[:args.max_cands]
        srcs = [torch.randn(*s, generator=g).to(DEV) for s in spec["src_shapes"]]
        out = torch.empty(*spec["out_shape"], device=DEV)
        res = torch.randn(*spec["out_shape"], generator=g).to(DEV) if spec["residual"] else None
        weight = torch.randn(cout, cin, kh, kw, generator=g) * 0.05
        bias = torch.randn(cout, generator=g)
        phase_w = [torch.randn(cout, cin, ph[4], ph[5], generator=g) * 0.05 for ph in spec["phases"]] if nph > 1 else None
        best = None
        rows = []
        for cd in keep:
            sched = (cd["mb"], cd["nb"], cd["split_k"], cd["ck"], cd["waves"], cd.get("kws", 0))
            try:
                p, fn = build_candidate(spec, sched, (srcs, out, res, weight if nph == 1 else None, bias, phase_w), c.get("bf16", False))
                t = time_op(fn, reps=args.reps) if args.streams <= 1 else time_op_streams(fn, streams, reps=args.reps)
            except RuntimeError as e:
                rows.append((sched, None, str(e)[:60]))
                continue
            rows.append((sched, t, cd["wgs"]))
            if best is None or t < best[0]:
                best = (t, sched)
            del p, fn
        # second pass: the leaders of the (noisy, few-repetition) sweep again, interleaved and with many repetitions
        finalists = [r for r in sorted((r for r in rows if r[1]), key=lambda r: r[1])[:args.finalists]]
        if len(finalists) > 1 and args.final_reps > 0:
            built = []
            for sched, _, _ in finalists:
                p, fn = build_candidate(spec, sched, (srcs, out, res, weight if nph == 1 else None, bias, phase_w), c.get("bf16", False))
                built.append((sched, p, fn, []))
            for _ in range(3):
                for sched, p, fn, ts in built:
                    ts.append(time_op(fn, reps=args.final_reps, warm=1))
            scored = sorted((sorted(ts)[1], sched) for sched, p, fn, ts in built)      # median of three rounds
            best = scored[0]
            del built
        nosplit = None
        if args.prefer_nosplit is not None and best[1][2] > 1:
            # the leaders without split_k, re-timed next to the overall best in the same interleaved way
            ns = [r for r in sorted((r for r in rows if r[1] and r[0][2] == 1), key=lambda r: r[1])[:3]]
            if ns:
                built = []
                for sched in [best[1]] + [r[0] for r in ns]:
                    p, fn = build_candidate(spec, sched, (srcs, out, res, weight if nph == 1 else None, bias, phase_w), c.get("bf16", False))
                    built.append((sched, p, fn, []))
                for _ in range(3):
                    for sched, p, fn, ts in built:
                        ts.append(time_op(fn, reps=max(args.final_reps, 20), warm=1))
                t_best = sorted(built[0][3])[1]
                t_ns, s_ns = min((sorted(ts)[1], sched) for sched, p, fn, ts in built[1:])
                nosplit = (t_ns, s_ns, t_best)
                if t_ns <= t_best * (1.0 + args.prefer_nosplit / 100.0):
                    best = (t_ns, s_ns)
                del built
        table[c["sig"]] = list(best[1])
        tf = 2 * c["macs"] / best[0] / 1e12
        report.append(dict(name=c["name"], sig=c["sig"], best=best[1], us=best[0] * 1e6, tflops=tf,
                           tried=[(list(s), (t * 1e6 if t else None), w) for s, t, w in sorted(rows, key=lambda r: r[1] or 1e9)[:8]]))
        print(f"{c['name']:26s} best={best[1]} {best[0]*1e6:8.1f} us {tf:6.1f} TF  ({len(keep)} tried)"
              + (f"  [no split_k: {nosplit[1]} {nosplit[0]*1e6:.1f} us vs {nosplit[2]*1e6:.1f} us]" if nosplit else ""), flush=True)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    total = sum(r["us"] for r in report)
    print(f"sum of best conv times (unique signatures): {total/1e3:.3f} ms; tuning took {time.time()-t_start:.1f} s")
    if args.report:
        os.makedirs(os.path.dirname(os.path.abspath(args.report)), exist_ok=True)
        json.dump(report, open(args.report, "w"), indent=1)


if __name__ == "__main__":
    main()
