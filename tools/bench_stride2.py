#!/usr/bin/env python
"""The stride-2 ConvReLU2 pairs of the DepthModule (enc stages 1-3: 7 / 5 / 5 taps; reference model/layers.py:289-314, monorec_model.py:489-501)
on the stride-1 Cook-Toom kernel over [even | odd] views (Plan._conv_relu2_stride2: F(4,4) for 7 taps, F(4,3) for 5) next to the two direct
MFMA launches with their tuned schedules: max |difference| of the pair's output and HIP-event times of the PAIR; --emit merges the fastest
choice per shape into the measured table (keys s2k<taps>_co<cout>_ci<cin>_o<h>x<w>_b<batch>: 0 = direct, 10 * blocks(k x 1 half) + blocks(1 x k half)).

    python tools/bench_stride2.py [--batch 1 --frames 2 --depths 32 --height 256 --width 512] [--emit monorec_amd/tuned_winograd.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import engine, synth                              # noqa: E402
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
    ap.add_argument("--margin", type=float, default=0.97, help="a form must beat margin x the direct pair to enter the table")
    a = ap.parse_args()
    m = MonoRecModel(cv_depth_steps=a.depths)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    for key in [k for k in engine.WINOGRAD if k.startswith("s2k")]:
        del engine.WINOGRAD[key]
    ref_plan = engine.Plan(sd, a.batch, a.height, a.width, a.frames, a.depths, (0.33, 0.0025), "cpu")
    log = {c["name"]: c for c in ref_plan.conv_log}
    g = torch.Generator().manual_seed(0)
    table, tot_d, tot_b = {}, 0.0, 0.0
    for i in (1, 2, 3):
        cy, cx = log[f"depth.enc{i}.0.conv_y"], log[f"depth.enc{i}.0.conv_x"]
        k = max(cy["k"])
        n, cin, h, w = cy["spec"]["src_shapes"][0]
        cm, co = cy["cout"], cx["cout"]
        h2, w2 = h // 2, w // 2
        sig = engine.stride2_signature(k, cm, cin, h2, w2, n)
        pre = f"depth_module.enc.{i}.0"
        x = torch.randn(n, cin, h, w, generator=g).to(DEV)
        row = {"name": f"depth.enc{i}.0", "sig": sig, "taps": k, "cin": cin, "cmid": cm, "cout": co, "in_hw": [h, w], "n": n,
               "ref_gmac": round((cy["ref_macs"] + cx["ref_macs"]) / 1e9, 3)}
        outs = {}
        blocks_y = [b for b in (1, 2, 3, 4) if 16 * b < 2 * cm or b == 1]
        blocks_x = [b for b in (1, 2, 3, 4) if 16 * b < 2 * co or b == 1]
        codes = [0] + [10 * by + bx for by in blocks_y for bx in [0] + blocks_x]        # bx = 0: only the k x 1 half on the Cook-Toom kernel
        best_y = {}
        for code in codes:
            engine.WINOGRAD[sig] = code
            plan = engine.Plan.bare(DEV, state=sd)
            plan.winograd = True
            mid = torch.full((n, cm, h2, w), float("nan"), device=DEV)
            out = torch.full((n, co, h2, w2), float("nan"), device=DEV)
            plan.conv_relu2("main", f"depth.enc{i}.0", [x], pre, mid, out, stride=2)
            plan.finalize()
            assert [bool(c.get("stride2")) for c in plan.conv_log] == [bool(code), bool(code % 10)], code
            fns = [f for _, f in plan.stages["main"]]
            st = torch.cuda.current_stream().cuda_stream
            for f in fns:
                f(st)
            torch.cuda.synchronize()
            outs[code] = out
            ty, tx = timed(fns[0]), timed(fns[1])
            tp = timed(lambda s_: (fns[0](s_), fns[1](s_)))
            tag = "direct" if code == 0 else f"s{code}"
            row[f"{tag}_us"] = [round(ty, 1), round(tx, 1), round(tp, 1)]
            if code:
                row[f"{tag}_maxdiff"] = float((out - outs[0]).abs().max())
        # the two halves are independent launches: the best pair = best k x 1 half + best 1 x k half (timed alone), confirmed by the pair timing
        best, tb = 0, a.margin * row["direct_us"][2]
        for code in codes[1:]:
            if row[f"s{code}_us"][2] < tb:
                best, tb = code, row[f"s{code}_us"][2]
        row["best"] = best
        row["algorithmic_tflops_best"] = round(2 * (cy["ref_macs"] + cx["ref_macs"]) / (min(tb, row["direct_us"][2]) * 1e-6) / 1e12, 1)
        row["algorithmic_tflops_direct"] = round(2 * (cy["ref_macs"] + cx["ref_macs"]) / (row["direct_us"][2] * 1e-6) / 1e12, 1)
        table[sig] = best
        tot_d += row["direct_us"][2]
        tot_b += min(row["direct_us"][2], tb if best else 1e9)
        print(json.dumps(row), flush=True)
    print(json.dumps({"direct_total_us": round(tot_d, 1), "best_total_us": round(tot_b, 1)}))
    if a.emit:
        old = json.load(open(a.emit)) if os.path.exists(a.emit) else {}
        old.update(table)
        with open(a.emit, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
