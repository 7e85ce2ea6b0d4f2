#!/usr/bin/env python
"""The 1-D Winograd kernels - F(2,3) (mr_conv1d3_winograd_f32) and the Cook-Toom forms F(4,3) / F(2,7) / F(4,7) (mr_conv1d_cooktoom_f32) -
next to the direct MFMA kernel (mr_conv2d_f32 with its tuned schedule) on the k x 1 / 1 x k stride-1 layers of a plan (k = 3, 7): max
|difference| and HIP-event times; --emit merges the fastest choice per layer shape into the measured table (keys x_<sig> / y_<sig> for 3
taps, x7_ / y7_ for 7: 0 direct, 1..4 = F(2,3) with 16 x that many output channels per workgroup, 10 m + mbw = F(m, taps)).

    python tools/bench_wino1d.py [--batch 1 --frames 2 --depths 32 --height 256 --width 512] [--emit monorec_amd/tuned_winograd.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import _lib, engine, synth                        # noqa: E402
from monorec_amd.model import MonoRecModel                         # noqa: E402
from tools.bench_wino import timed                                  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--emit", default=None)
    ap.add_argument("--no-upconv", action="store_true", help="skip the Upconv layers (their table does not change)")
    ap.add_argument("--taps", default="3,7", help="which filter lengths to measure")
    a = ap.parse_args()
    diagnostic = _lib.load().has_diagnostic_forms      # F(2,7) lives in the diagnostic build only (round 4)
    m = MonoRecModel(cv_depth_steps=a.depths)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    ref_plan = engine.Plan(sd, a.batch, a.height, a.width, a.frames, a.depths, (0.33, 0.0025), "cpu", winograd=False)
    g = torch.Generator().manual_seed(0)
    table, tot_d, tot_b, tot_old = {}, 0.0, 0.0, 0.0
    for c in ref_plan.conv_log:
        sp = c["spec"]
        taps = max(c["k"])
        if (min(c["k"]) != 1 or str(taps) not in a.taps.split(",") or taps not in (3, 7) or tuple(sp["stride"]) != (1, 1) or c["phases"] != 1 or
                sp["in_mode"] != 0 or sp["tf"] != 0):
            continue
        axis = 0 if c["k"][0] == 1 else 1
        srcs = [torch.randn(*s, generator=g).to(DEV) for s in sp["src_shapes"]]
        cout, cin = sp["w_shape"][0], sp["w_shape"][1]
        w = torch.randn(cout, cin, *c["k"], generator=g) * (1.0 / (taps * cin) ** 0.5)
        bias = torch.randn(cout, generator=g) * 0.1
        sc = [int(s.shape[1]) for s in srcs]
        sig = ("x", "y")[axis] + ("" if taps == 3 else str(taps)) + "_" + engine.winograd_signature(cout, sc, sp["grid"][0], sp["grid"][1], sp["out_shape"][0])
        row = {"name": c["name"], "sig": sig, "cin": cin, "cout": cout, "hw": list(sp["grid"]), "n": sp["out_shape"][0]}
        outs = {}
        codes = (0, 1, 2, 3, 4, 41, 42, 43, 44) if taps == 3 else ((0, 21, 22, 23, 24, 41, 42, 43) if diagnostic else (0, 41, 42, 43))
        for code in codes:
            if code and 16 * (code % 10) >= 2 * cout and code % 10 > 1:
                continue
            engine.WINOGRAD[sig] = code
            plan = engine.Plan.bare(DEV)
            plan.winograd = True
            out = torch.full(sp["out_shape"], float("nan"), device=DEV)
            plan.conv("main", c["name"], srcs, w, bias, out, stride=(1, 1), pad=sp["pad"], grid=sp["grid"], act=sp["act"], p0=sp["p0"])
            plan.finalize()
            assert bool(plan.conv_log[0].get("winograd")) == bool(code)
            fn = plan.stages["main"][0][1]
            fn(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs[code] = out
            row["direct_us" if code == 0 else f"wino{code}_us"] = round(timed(fn), 1)
            if code:
                row[f"wino{code}_maxdiff"] = float((out - outs[0]).abs().max())
        best, tb = 0, 0.97 * row["direct_us"]
        for code in codes[1:]:
            if f"wino{code}_us" in row and row[f"wino{code}_us"] < tb:
                best, tb = code, row[f"wino{code}_us"]
        row["best"] = best
        table[sig] = best
        tot_d += row["direct_us"]
        tot_b += min(row["direct_us"], tb if best else 1e9)
        tot_old += min([row["direct_us"] / 0.97] + [row[f"wino{code_}_us"] for code_ in (1, 2, 3, 4) if f"wino{code_}_us" in row]) if taps == 3 else row["direct_us"]
        print(json.dumps(row), flush=True)
    print(json.dumps({"direct_total_us": round(tot_d, 1), "best_of_direct_and_f23_total_us": round(tot_old, 1), "best_of_all_total_us": round(tot_b, 1)}))
    if a.no_upconv:
        ref_plan.conv_log.clear()
    # ---- layers.Upconv: the four parity phases on the direct kernel (9 multiplies per 2x2 block) vs the 4-multiply kernel -------------
    tot_d = tot_b = 0.0
    for c in ref_plan.conv_log:
        sp = c["spec"]
        if c["phases"] != 4 or not c["name"].startswith("mask.dec") or not c["name"].endswith(".0"):
            continue
        srcs = [torch.randn(*s_, generator=g).to(DEV) for s_ in sp["src_shapes"]]
        cout, cin = sp["w_shape"][0], sp["w_shape"][1]
        w = torch.randn(cout, cin, 2, 2, generator=g) * (1.0 / (4.0 * cin) ** 0.5)
        bias = torch.randn(cout, generator=g) * 0.1
        sc = [int(s_.shape[1]) for s_ in srcs]
        hs, ws, n = srcs[0].shape[2], srcs[0].shape[3], srcs[0].shape[0]
        sig = "u_" + engine.winograd_signature(cout, sc, hs, ws, n)
        row = {"name": c["name"], "sig": sig, "cin": cin, "cout": cout, "hw": [hs, ws], "n": n}
        outs = {}
        for code in (0, 1, 2):
            engine.WINOGRAD[sig] = code
            plan = engine.Plan.bare(DEV, state={"x.weight": w, "x.bias": bias})
            plan.winograd = True
            out = torch.full((n, cout, 2 * hs, 2 * ws), float("nan"), device=DEV)
            plan.upconv("main", c["name"], srcs, "x.weight", "x.bias", out)
            plan.finalize()
            assert bool(plan.conv_log[0].get("winograd")) == bool(code)
            fn = plan.stages["main"][0][1]
            fn(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs[code] = out
            row["direct_us" if code == 0 else f"wino{code}_us"] = round(timed(fn), 1)
            if code:
                row[f"wino{code}_maxdiff"] = float((out - outs[0]).abs().max())
        best, tb = 0, 0.97 * row["direct_us"]
        for code in (1, 2):
            if row[f"wino{code}_us"] < tb:
                best, tb = code, row[f"wino{code}_us"]
        row["best"] = best
        table[sig] = best
        tot_d += row["direct_us"]
        tot_b += min(row["direct_us"], tb if best else 1e9)
        print(json.dumps(row), flush=True)
    print(json.dumps({"upconv_direct_total_us": round(tot_d, 1), "upconv_best_total_us": round(tot_b, 1)}))
    if a.emit:
        old = json.load(open(a.emit)) if os.path.exists(a.emit) else {}
        old.update(table)
        with open(a.emit, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
