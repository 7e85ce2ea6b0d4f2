#!/usr/bin/env python
"""Time mr_depth_heads_f32 / mr_mask_classifier_f32 on the GPU box (HIP events on the launch stream): every head of a decoder
on its own and all four together.

    python tools/bench_heads.py [--batch 1] [--height 256] [--width 512] [--depths 32]
    MR_HIP_LIBRARY=monorec_amd/libmonorec_hip_timeline.so MR_HEADS_QUAD_MIN=1 python tools/bench_heads.py       # every head in quad mode;  MR_HEADS_QUAD_MIN=1000000000: all in pixel mode
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import _lib                                          # noqa: E402


def timed(fn, iters=200):
    s = torch.cuda.current_stream()
    for _ in range(10):
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
    return round(best, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--depths", type=int, default=32)
    a = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    b, h, w = a.batch, a.height, a.width
    shapes = [(b, 256, h // 8, w // 8), (b, 128, h // 4, w // 4), (b, 64, h // 2, w // 2), (b, 24, h, w)]
    keep, descs = [], (_lib.HeadDesc * 4)()
    for i, (bb, c, hh, ww) in enumerate(shapes):
        x, wt, bias, out = torch.randn(bb, c, hh, ww, device=dev), torch.randn(1, c, 3, 3, device=dev) * 0.02, torch.zeros(1, device=dev), torch.empty(bb, 1, hh, ww, device=dev)
        keep += [x, wt, bias, out]
        descs[i].src, descs[i].weight, descs[i].bias, descs[i].dst = x.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr()
        descs[i].batch, descs[i].channels, descs[i].height, descs[i].width = bb, c, hh, ww
    out = {"shape": [b, h, w], "MR_HEADS_QUAD_MIN": os.environ.get("MR_HEADS_QUAD_MIN")}
    for i in range(4):
        one = (_lib.HeadDesc * 1)(descs[i])
        out[f"head{i}_us"] = timed(lambda s, one=one: _lib.check(lib.mr_depth_heads_f32(one, 1, 0.0025, 0.33, s)))
    out["all_four_us"] = timed(lambda s: _lib.check(lib.mr_depth_heads_f32(descs, 4, 0.0025, 0.33, s)))
    feat, cw, cb = torch.randn(b, 48, h, w, device=dev), torch.randn(48, device=dev) * 0.1, torch.zeros(1, device=dev)
    mask, cv = torch.empty(b, 1, h, w, device=dev), torch.randn(b, a.depths, h, w, device=dev)
    out["classifier_apply_us"] = timed(lambda s: _lib.check(lib.mr_mask_classifier_f32(feat.data_ptr(), cw.data_ptr(), cb.data_ptr(), b, 48, h * w,
                                                                                      mask.data_ptr(), cv.data_ptr(), a.depths, s)))
    out["classifier_only_us"] = timed(lambda s: _lib.check(lib.mr_mask_classifier_f32(feat.data_ptr(), cw.data_ptr(), cb.data_ptr(), b, 48, h * w,
                                                                                     mask.data_ptr(), None, a.depths, s)))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
