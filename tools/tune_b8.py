#!/usr/bin/env python
"""Measured (MB, NB, waves) per launch of the bf16 / B8 convolution kernel (csrc/conv_b8.hip; `hip_bf16=True` plans): every launch signature of a plan
is rebuilt in isolation on random data and timed with HIP events under each schedule the kernel can launch; a schedule enters the table
(monorec_amd/tuned_b8.json, read by engine.Plan.conv_b8 ahead of the rule Plan.b8_schedule) only where it beats the rule's choice by --margin.

    python tools/tune_b8.py --height 512 --width 1024 --frames 4 --depths 48 [--emit monorec_amd/tuned_b8.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import engine, synth                              # noqa: E402
from monorec_amd._lib import ACT_NONE                               # noqa: E402
from monorec_amd.model import MonoRecModel                         # noqa: E402

DEV = "cuda:0"
TILES = ((8, 4), (8, 2), (8, 1), (4, 2), (4, 1))                    # (waves, pixel blocks per wave)


def build(c, sched, g):
    """One plan holding the launch of conv-log entry `c` under `sched`; None if the kernel cannot launch it."""
    spec, cout, cin = c["spec"], c["cout"], c["cin"]
    kh, kw = spec["w_shape"][2], spec["w_shape"][3]
    h, w = spec["grid"]
    n = spec["src_shapes"][0][0]
    sd = {}
    four = c["phases"] == 4
    upconv = four and spec["act"] == ACT_NONE
    if four and upconv:
        sd["t.weight"] = torch.randn(cout, cin, 2, 2, generator=g) * (0.5 / cin ** 0.5)
        sd["t.bias"] = torch.zeros(cout)
    elif four:
        sd["t.conv2d_t.weight"] = torch.randn(cin, cout, 4, 4, generator=g) * (0.5 / cin ** 0.5)
        sd["t.conv2d_t.bias"] = torch.zeros(cout)
    plan = engine.Plan.bare(DEV, state=sd, schedule_override={"t": sched}, bf16=1)
    srcs = []
    for i, ((sn, sc, sh_, sw_), lay) in enumerate(zip(spec["src_shapes"], spec["src_layouts"])):
        if lay:
            t = plan.alloc_b8(f"s{i}", sn, sc, sh_, sw_)
            t.normal_()
        else:
            t = torch.randn(sn, sc, sh_, sw_, generator=g).to(DEV)
        srcs.append(t)
    oh, ow = (2 * h, 2 * w) if four else (h, w)
    out = plan.alloc_b8("o", n, cout, oh, ow) if spec["out_layout"] else torch.empty(n, cout, oh, ow, device=DEV)
    try:
        if four and upconv:
            plan.upconv_b8("main", "t", srcs, "t.weight", "t.bias", out)
        elif four:
            plan.refine_b8("main", "t", srcs, "t", out)
        else:
            wt = torch.randn(cout, cin, kh, kw, generator=g) * (1.0 / (kh * kw * cin) ** 0.5)
            plan.conv_b8("main", "t", srcs, wt, torch.zeros(cout), out, stride=spec["stride"], pad=spec["pad"], grid=spec["grid"], act=spec["act"], p0=spec["p0"],
                         out_step=spec["out_step"])
    except (RuntimeError, AssertionError, ValueError):
        return None
    plan.finalize()
    return plan


def timed(plan, reps, rounds=3):
    s = torch.cuda.current_stream()
    for _ in range(3):
        plan.run_stage("main", s.cuda_stream)
    torch.cuda.synchronize()
    best = None
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.run_stage("main", s.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / reps
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--depths", type=int, default=48)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--margin", type=float, default=0.97, help="a schedule must beat margin x the rule's time to enter the table")
    ap.add_argument("--emit", default=None)
    a = ap.parse_args()
    for k in list(engine.B8_SCHEDULES):
        del engine.B8_SCHEDULES[k]                                   # the rule's choice is the reference point
    m = MonoRecModel(cv_depth_steps=a.depths)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    ref = engine.Plan(sd, a.batch, a.height, a.width, a.frames, a.depths, (0.33, 0.0025), "cpu", bf16=1)
    g = torch.Generator().manual_seed(0)
    seen, table, tot_rule, tot_best = set(), {}, 0.0, 0.0
    for c in ref.conv_log:
        if not c.get("b8"):
            continue
        key = c["sig"] + f"_f{int(c['f32_source'])}"
        if key in seen:
            continue
        seen.add(key)
        rule = (c["mb"], c["nb"], c["waves"])
        cb16 = (c["cout"] + 15) // 16
        mbs = sorted({rule[0]} | {m_ for m_ in (1, 2, 3, 4) if m_ <= cb16 and (cb16 % m_ == 0 or m_ == rule[0])}, reverse=True)
        row = {"name": c["name"], "sig": key, "rule": list(rule), "times_us": {}}
        for mb in mbs:
            for waves, nb in TILES:
                sched = (mb, nb, waves)
                plan = build(c, sched, g)
                if plan is None:
                    continue
                try:
                    row["times_us"]["%d,%d,%d" % sched] = round(timed(plan, a.reps), 1)
                except RuntimeError:                                  # a combination the launcher has no instantiation for
                    pass
                del plan
        t_rule = row["times_us"].get("%d,%d,%d" % rule)
        if t_rule is None:
            print(json.dumps(dict(row, error="the rule's schedule did not launch")), flush=True)
            continue
        best_s, best_t = min(row["times_us"].items(), key=lambda kv: kv[1])
        row["best"], row["rule_us"], row["best_us"] = best_s, t_rule, best_t
        n_same = sum(1 for c2 in ref.conv_log if c2.get("b8") and c2["sig"] + f"_f{int(c2['f32_source'])}" == key)
        tot_rule += n_same * t_rule
        if best_t < a.margin * t_rule:
            table[key] = [int(x) for x in best_s.split(",")]
            tot_best += n_same * best_t
        else:
            tot_best += n_same * t_rule
        print(json.dumps(row), flush=True)
    print(json.dumps({"launches": sum(1 for c in ref.conv_log if c.get("b8")), "signatures": len(seen), "rule_total_us": round(tot_rule, 1),
                      "table_total_us": round(tot_best, 1), "entries": len(table)}))
    if a.emit:
        old = json.load(open(a.emit)) if os.path.exists(a.emit) else {}
        old.update(table)
        with open(a.emit, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
