#!/usr/bin/env python
"""Winograd F(2x2,3x3) kernel (mr_conv3x3_winograd_f32) next to the direct MFMA kernel (mr_conv2d_f32 with its tuned schedule) on
the 3x3 stride-1 layers of a plan: max |difference| between the two and HIP-event times of both.

    python tools/bench_wino.py [--batch 1 --frames 2 --depths 32 --height 256 --width 512] [--only mask.enc0]
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

DEV = "cuda:0"


def timed(fn, iters=50):
    s = torch.cuda.current_stream()
    for _ in range(5):
        fn(s.cuda_stream)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters):
            fn(s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / iters
        best = t if best is None else min(best, t)
    return best


def wino_launch(lib, srcs, weight, bias, out, act, p0, mbw, residual=None, variant=0):
    sc = [int(s.shape[1]) for s in srcs]
    arr = (ctypes.c_int32 * len(sc))(*sc)
    if variant == 4:                                   # F(4x4,3x3), positions split over two waves: csrc/conv_wino44s.hip
        n = lib.mr_wino44s_packed_weight_floats(weight.shape[0], arr, len(sc))
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(lib.mr_wino44s_pack_weights_f32(weight.contiguous().data_ptr(), weight.shape[0], arr, len(sc), packed.data_ptr()), "pack")
    elif variant in (3, 5):                            # F(4x4,3x3): csrc/conv_wino44.hip / conv_wino44w.hip (one wave per SIMD)
        n = lib.mr_wino44_packed_weight_floats(weight.shape[0], arr, len(sc))
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(lib.mr_wino44_pack_weights_f32(weight.contiguous().data_ptr(), weight.shape[0], arr, len(sc), packed.data_ptr()), "pack")
    elif variant == 2:
        n = lib.mr_wino_packed_weight_floats_tail(weight.shape[0], arr, len(sc))
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(lib.mr_wino_pack_weights_tail_f32(weight.contiguous().data_ptr(), weight.shape[0], arr, len(sc), packed.data_ptr()), "pack")
    else:
        n = lib.mr_wino_packed_weight_floats(weight.shape[0], arr, len(sc), mbw)
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(lib.mr_wino_pack_weights_f32(weight.contiguous().data_ptr(), weight.shape[0], arr, len(sc), mbw, packed.data_ptr()), "pack")
    d = _lib.WinoDesc()
    for i, s in enumerate(srcs):
        d.src[i], d.src_channels[i] = s.data_ptr(), sc[i]
    d.num_src, d.batch, d.height, d.width = len(srcs), srcs[0].shape[0], srcs[0].shape[2], srcs[0].shape[3]
    d.dst, d.out_channels = out.data_ptr(), weight.shape[0]
    pk, bs = packed.to(DEV), (bias.to(DEV) if bias is not None else None)
    d.packed_weights, d.bias = pk.data_ptr(), (bs.data_ptr() if bs is not None else None)
    d.residual = residual.data_ptr() if residual is not None else None
    d.activation, d.act_p0, d.cout_blocks_per_wave, d.variant = act, p0, mbw, variant
    keep = (pk, bs, d)
    if variant == 4:
        return (lambda stream: _lib.check(lib.mr_conv3x3_winograd44s_f32(ctypes.byref(d), stream), "wino44s")), keep
    if variant == 5:
        return (lambda stream: _lib.check(lib.mr_conv3x3_winograd44w_f32(ctypes.byref(d), stream), "wino44w")), keep
    if variant == 3:
        return (lambda stream: _lib.check(lib.mr_conv3x3_winograd44_f32(ctypes.byref(d), stream), "wino44")), keep
    return (lambda stream: _lib.check(lib.mr_conv3x3_winograd_f32(ctypes.byref(d), stream), "wino")), keep


CODES = (1, 2, 11, 12, 21, 31, 41, 51)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--only", default=None)
    ap.add_argument("--codes", default=None, help="comma-separated subset of the kernel codes to time (default: all)")
    ap.add_argument("--min-pixels", type=int, default=0, help="skip layers with fewer output pixels per image (a quick pass over the big layers)")
    ap.add_argument("--emit", default=None, help="write {signature: 0 | 1 | 2 | 11 | 12} (fastest kernel per layer; Winograd must win by 3 %%; + 10 = input "
                                                 "transform in registers) to this JSON file; entries already in the file for other shapes are kept")
    a = ap.parse_args()
    lib = _lib.load()
    codes = CODES if a.codes is None else tuple(int(c) for c in a.codes.split(","))
    m = MonoRecModel(cv_depth_steps=a.depths)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    ref_plan = engine.Plan(sd, a.batch, a.height, a.width, a.frames, a.depths, (0.33, 0.0025), "cpu", winograd=False)
    g = torch.Generator().manual_seed(0)
    rows = []
    for c in ref_plan.conv_log:
        sp = c["spec"]
        if tuple(c["k"]) != (3, 3) or tuple(sp["stride"]) != (1, 1) or c["phases"] != 1 or sp["in_mode"] != 0 or sp["tf"] != 0:
            continue
        if a.only and a.only not in c["name"]:
            continue
        if sp["grid"][0] * sp["grid"][1] < a.min_pixels:
            continue
        srcs = [torch.randn(*s, generator=g).to(DEV) for s in sp["src_shapes"]]
        cout, cin = sp["w_shape"][0], sp["w_shape"][1]
        w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3.0 * cin ** 0.5))
        bias = torch.randn(cout, generator=g) * 0.1
        res = torch.randn(*sp["out_shape"], generator=g).to(DEV) if sp["residual"] else None
        plan = engine.Plan.bare(DEV)
        plan.winograd = False                                             # the direct kernel, whatever the measured table says
        out_d = torch.empty(*sp["out_shape"], device=DEV)
        plan.conv("main", c["name"], srcs, w, bias, out_d, stride=(1, 1), pad=(1, 1), grid=sp["grid"], act=sp["act"], p0=sp["p0"], residual=res)
        plan.finalize()
        direct = plan.stages["main"][0][1]
        row = {"name": c["name"], "cin": cin, "cout": cout, "hw": list(sp["grid"]), "n": sp["out_shape"][0], "sched": [c["mb"], c["nb"], c["split_k"], c["ck"], c["waves"]],
               "direct_us": round(timed(direct), 1)}
        for code in codes:                   # cout blocks per wave, + 10: input transform in registers, + 20: ... with tail workgroups, 31: F(4x4,3x3), 41: F(4x4,3x3) split over two waves
            mbw, variant = code % 10, code // 10
            if mbw == 2 and cout <= 32:
                continue
            if variant == 2 and not (0 < cout % 32 <= 16):
                continue
            out_w = torch.full(sp["out_shape"], float("nan"), device=DEV)
            fn, keep = wino_launch(lib, srcs, w, bias, out_w, sp["act"], sp["p0"], mbw, res, variant)
            fn(torch.cuda.current_stream().cuda_stream)
            direct(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            row[f"wino{code}_maxdiff"] = float((out_w - out_d).abs().max())
            row[f"wino{code}_us"] = round(timed(fn), 1)
        best, tb = 0, 0.97 * row["direct_us"]                # Winograd must win by 3 %
        for code in CODES:
            if f"wino{code}_us" in row and row[f"wino{code}_us"] < tb:
                best, tb = code, row[f"wino{code}_us"]
        row["sig"] = engine.winograd_signature(cout, [s_[1] for s_ in sp["src_shapes"]], sp["grid"][0], sp["grid"][1], sp["out_shape"][0])
        row["best"] = best
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.emit:
        os.makedirs(os.path.dirname(os.path.abspath(a.emit)), exist_ok=True)
        table = json.load(open(a.emit)) if os.path.exists(a.emit) else {}
        table.update({r["sig"]: r["best"] for r in rows})
        with open(a.emit, "w") as f:
            json.dump(table, f, indent=0, sort_keys=True)
    tot_d = sum(r["direct_us"] for r in rows)
    tot_w = sum(min([r["direct_us"]] + [r.get(f"wino{c}_us", 1e9) for c in CODES]) for r in rows)
    print(json.dumps({"layers": len(rows), "direct_total_us": round(tot_d, 1), "best_of_both_total_us": round(tot_w, 1)}))


if __name__ == "__main__":
    main()
