#!/usr/bin/env python
"""The F(2x2,2x2) transposed-convolution kernel (mr_convt4x4s2_winograd_f32) next to the direct MFMA kernel (four parity phases of
mr_conv2d_f32 with its tuned schedule) on the Refine layers of a plan: max |difference| and HIP-event times.

    python tools/bench_wino_t.py [--batch 1 --frames 2 --depths 32 --height 256 --width 512]
"""
import argparse
import ctypes
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
    ap.add_argument("--emit", default=None, help="merge {t_<signature>: 0 | mbw | 10 + mbw} (fastest; Winograd must win by 3 %%) into this JSON table")
    a = ap.parse_args()
    table = {}
    lib = _lib.load()
    m = MonoRecModel(cv_depth_steps=a.depths)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    ref_plan = engine.Plan(sd, a.batch, a.height, a.width, a.frames, a.depths, (0.33, 0.0025), "cpu", winograd=False)
    g = torch.Generator().manual_seed(0)
    tot_d = tot_b = 0.0
    for c in ref_plan.conv_log:
        sp = c["spec"]
        if c["phases"] != 4 or tuple(c["k"]) != (2, 2) or not c["name"].startswith("depth.dec"):
            continue
        srcs = [torch.randn(*s, generator=g).to(DEV) for s in sp["src_shapes"]]
        cout, cin = sp["w_shape"][0], sp["w_shape"][1]
        wt = torch.randn(cin, cout, 4, 4, generator=g) * (1.0 / (2.0 * cin ** 0.5))
        bias = torch.randn(cout, generator=g) * 0.1
        plan = engine.Plan.bare(DEV, state={"x.conv2d_t.weight": wt, "x.conv2d_t.bias": bias})
        plan.winograd = False
        out_d = torch.empty(*sp["out_shape"], device=DEV)
        plan.schedule_override = {}
        plan.refine("main", c["name"], srcs, "x", out_d)
        plan.finalize()
        direct = plan.stages["main"][0][1]
        row = {"name": c["name"], "cin": cin, "cout": cout, "hw": list(sp["grid"]), "n": sp["out_shape"][0],
               "sched": [plan.conv_log[0]["mb"], plan.conv_log[0]["nb"], plan.conv_log[0]["split_k"], plan.conv_log[0]["ck"], plan.conv_log[0]["waves"]],
               "direct_us": round(timed(direct), 1)}
        sc = [int(s.shape[1]) for s in srcs]
        arr = (ctypes.c_int32 * len(sc))(*sc)
        best, best_code = row["direct_us"], 0
        for code in (1, 2, 4, 11, 12, 14, 21):               # output-channel blocks per wave; + 10: input transform in registers; + 20: ... with tail workgroups
            mbw, variant = code % 10, code // 10
            if 32 * mbw >= 2 * cout and mbw > 1:
                continue
            if variant == 2:
                if not 0 < cout % 32 <= 16:
                    continue
                n = lib.mr_wino_t_packed_weight_floats_tail(cout, arr, len(sc))
                packed = torch.empty(n)
                _lib.check(lib.mr_wino_t_pack_weights_tail_f32(wt.data_ptr(), cout, arr, len(sc), packed.data_ptr()), "pack")
            else:
                n = lib.mr_wino_t_packed_weight_floats(cout, arr, len(sc), mbw)
                packed = torch.empty(n)
                _lib.check(lib.mr_wino_t_pack_weights_f32(wt.data_ptr(), cout, arr, len(sc), mbw, packed.data_ptr()), "pack")
            d = _lib.WinoDesc()
            for i, s in enumerate(srcs):
                d.src[i], d.src_channels[i] = s.data_ptr(), sc[i]
            out_w = torch.full(sp["out_shape"], float("nan"), device=DEV)
            pk, bs = packed.to(DEV), bias.to(DEV)
            d.num_src, d.batch, d.height, d.width = len(srcs), srcs[0].shape[0], srcs[0].shape[2], srcs[0].shape[3]
            d.dst, d.out_channels, d.packed_weights, d.bias = out_w.data_ptr(), cout, pk.data_ptr(), bs.data_ptr()
            d.activation, d.act_p0, d.cout_blocks_per_wave, d.variant = sp["act"], sp["p0"], mbw, variant
            fn = lambda stream, d=d: _lib.check(lib.mr_convt4x4s2_winograd_f32(ctypes.byref(d), stream), "wino_t")   # noqa: E731
            fn(torch.cuda.current_stream().cuda_stream)
            direct(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            row[f"wino{code}_maxdiff"] = float((out_w - out_d).abs().max())
            row[f"wino{code}_us"] = round(timed(fn), 1)
            if row[f"wino{code}_us"] < best and row[f"wino{code}_us"] < 0.97 * row["direct_us"]:
                best, best_code = row[f"wino{code}_us"], code
        tot_d += row["direct_us"]
        tot_b += best
        row["best"] = best_code
        table["t_" + engine.winograd_signature(cout, sc, srcs[0].shape[2], srcs[0].shape[3], srcs[0].shape[0])] = best_code
        print(json.dumps(row), flush=True)
    print(json.dumps({"direct_total_us": round(tot_d, 1), "best_of_both_total_us": round(tot_b, 1)}))
    if a.emit:
        old = json.load(open(a.emit)) if os.path.exists(a.emit) else {}
        old.update(table)
        with open(a.emit, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
