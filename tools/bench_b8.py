#!/usr/bin/env python
"""One layer of the bf16 / B8 path (csrc/conv_b8.hip) in isolation: time per launch by HIP events for a list of schedules, and - with the
diagnostic library (MR_HIP_LIBRARY=monorec_amd/libmonorec_hip_timeline.so) - the MR_B8_DBG ablations (1 no sweep, 2 no input staging,
4 no weight DMA, 8 no stores).

    python tools/bench_b8.py --layer enc0.1 [--scheds 3,4,8 3,2,8] [--batch 4]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import engine                       # noqa: E402
from monorec_amd._lib import ACT_LEAKY_RELU          # noqa: E402

# name -> (source channels, source layouts, cout, (kh, kw), (sh, sw), out layout, batch multiplier)
LAYERS = {
    "enc0.0": ((48,), (0,), 48, (3, 3), (1, 1), 1, 4),
    "enc0.1": ((48,), (1,), 48, (3, 3), (1, 1), 1, 4),
    "dec3.1": ((48, 64), (1, 1), 48, (3, 3), (1, 1), 1, 1),
    "enc0.0y": ((48, 3), (0, 0), 48, (7, 1), (1, 1), 1, 1),
    "enc0.1x": ((48,), (1,), 48, (1, 3), (1, 1), 1, 1),
    "dec3": ((64, 64, 64), (1, 0, 0), 48, (2, 2), (1, 1), 1, 1),          # depth.dec3 as ONE 2x2 phase on the 256x512 input (of four)
    "dec2.1": ((64, 64, 96), (1, 0, 1), 64, (3, 3), (1, 1), 1, 1),        # mask.dec2.1 at 256x512 (use --height 256 --width 512)
    "enc1.0": ((48,), (1,), 48, (3, 3), (1, 1), 1, 4),                    # mask.enc1.* at 256x512
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="enc0.1")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--scheds", nargs="*", default=["3,4,8", "3,2,8"])
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    srcs_c, lays, cout, (kh, kw), (sh, sw), olay, bm = LAYERS[args.layer]
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    h, w = args.height, args.width
    wt = torch.randn(cout, sum(srcs_c), kh, kw, generator=g) / math.sqrt(kh * kw * sum(srcs_c))
    bias = torch.zeros(cout)
    for sched in args.scheds:
        plan = engine.Plan.bare(dev, schedule_override={"t": tuple(int(v) for v in sched.split(","))}, bf16=1)      # mb,nb,waves
        srcs = []
        for c, lay in zip(srcs_c, lays):
            if lay:
                t = plan.alloc_b8(f"s{len(srcs)}", bm, c, h, w)
                t.normal_()
            else:
                t = torch.randn(bm, c, h, w, device=dev)
            srcs.append(t)
        oh, ow = math.ceil(h / sh), math.ceil(w / sw)
        out = plan.alloc_b8("o", bm, cout, oh, ow) if olay else torch.empty(bm, cout, oh, ow, device=dev)
        pt, _ = engine.same_pad(h, kh, sh)
        pl, _ = engine.same_pad(w, kw, sw)
        plan.conv_b8("main", "t", srcs, wt, bias, out, stride=(sh, sw), pad=(pt, pl), grid=(oh, ow), act=ACT_LEAKY_RELU, p0=0.1)
        plan.finalize()
        stream = torch.cuda.current_stream()
        for _ in range(3):
            plan.run_stage("main", stream.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            plan.run_stage("main", stream.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        log = plan.conv_log[0]
        print(f"{args.layer} sched {sched} dbg {os.environ.get('MR_B8_DBG', '0')}: {us:8.1f} us  {2 * log['macs'] / us / 1e6:7.1f} TF  wgs {log['wgs']} lds {log['lds']}", flush=True)


if __name__ == "__main__":
    main()
