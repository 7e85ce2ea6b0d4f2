#!/usr/bin/env python
"""Time the cost-volume entry points on the GPU box: the default path (marching sad kernel + register fusion kernel) next to
the round-1 LDS-tiled kernels (mr_cost_volume_tiled_f32), HIP events on the launch stream.

    python tools/bench_cv.py --batch 1 --height 256 --width 512 --frames 2 --depths 32 [--iters 200]
    MR_HIP_LIBRARY=monorec_amd/libmonorec_hip_timeline.so MR_CV_MARCH_TY=32 python tools/bench_cv.py ...      # force the row-segment length of the marching kernel (read once per process)
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorec_amd import _lib, synth                                  # noqa: E402
from monorec_amd.model import depth_hypotheses, host_geometry        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--impl", default="both", choices=("both", "march", "tiled"))
    a = ap.parse_args()
    dev = "cuda:0"
    lib = _lib.load()
    batch = synth.make_batch(a.batch, a.height, a.width, a.frames, seed=1)
    kf = batch["keyframe"].to(dev)
    frames = [f.to(dev).contiguous() for f in batch["frames"]]
    kinv, proj = host_geometry(batch["keyframe_intrinsics"], batch["keyframe_pose"], batch["intrinsics"], batch["poses"])
    kinv, proj = kinv.to(dev), proj.to(dev)
    depths = depth_hypotheses((0.33, 0.0025), a.depths).to(dev)
    b, d, h, w, nf = a.batch, a.depths, a.height, a.width, a.frames
    cv = torch.empty(b, d, h, w, device=dev)
    sf = [torch.empty(b, d, h, w, device=dev) for _ in range(nf)]
    fp = (ctypes.c_void_p * nf)(*[f.data_ptr() for f in frames])
    sp = (ctypes.c_void_p * nf)(*[s.data_ptr() for s in sf])
    cw = (ctypes.c_float * 3)(5 / 32, 16 / 32, 11 / 32)
    stream = torch.cuda.current_stream()

    def run(tiled):
        fn = lib.mr_cost_volume_tiled_f32 if tiled else lib.mr_cost_volume_mode_f32
        _lib.check(fn(kf.data_ptr(), fp, nf, kinv.data_ptr(), proj.data_ptr(), depths.data_ptr(), b, d, h, w, 10.0, cw, 1, None, 1,
                      cv.data_ptr(), sp, stream.cuda_stream), "cost volume")

    out = {"shape": [b, h, w, nf, d], "MR_CV_MARCH_TY": os.environ.get("MR_CV_MARCH_TY")}
    bytes_alg = 4.0 * b * h * w * (3 + d) * (1 + nf)
    for name, tiled in (("march", False), ("tiled", True)):
        if a.impl not in ("both", name):
            continue
        for _ in range(10):
            run(tiled)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.iters):
                run(tiled)
            e1.record(stream)
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / a.iters
            best = t if best is None else min(best, t)
        out[name + "_us"] = round(best, 2)
        out[name + "_GBps"] = round(bytes_alg / best / 1e3, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
