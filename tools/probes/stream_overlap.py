#!/usr/bin/env python
"""How much do two streams overlap on this GPU when the host is out of the way?  Stages of a c2 plan (native launch lists: a handful of
host calls per ~20 launches) enqueued N times on ONE stream, and N times on each of TWO streams at once.  efficiency = 2 T(one) / T(two
streams): 1.0 = the second stream's kernels ran for free inside the first one's gaps and idle CUs, 0.5 = they serialised.
(The buffers are shared between the copies - timing only.)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from monorec_amd import MonoRecModel, synth  # noqa: E402

dev = torch.device("cuda:0")
m = MonoRecModel(cv_depth_steps=32, hip_in_flight=2)
m.load_state_dict(synth.seeded_state_dict(m.state_dict()))
m = m.to(dev).eval()
b = synth.clone_batch(synth.make_batch(1, 256, 512, 2), dev)
with torch.no_grad():
    for _ in range(4):
        m.submit(dict(b)).synchronize()
plans = list(m._plans.values())
assert len(plans) >= 2, "two slots expected"
pa, pb = plans[0], plans[1]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(jobs, n=30):
    """jobs: [(plan, stage, stream)]; every job's stage is enqueued n times on its stream; wall time from first enqueue to all done."""
    for p, st, s in jobs:
        p.run_stage(st, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for p, st, s in jobs:
            p.run_stage(st, s.cuda_stream)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_host / n * 1e6


out = {}
for st in ("encoder", "cv", "main"):
    one, h1 = timed([(pa, st, sa)])
    two, h2 = timed([(pa, st, sa), (pb, st, sb)])
    out[st] = {"one_stream_us": round(one, 1), "two_streams_us": round(two, 1), "efficiency": round(2 * one / two, 3),
               "host_us_one": round(h1, 1), "host_us_two": round(h2, 1)}
enc, _ = timed([(pa, "encoder", sa)])
main, _ = timed([(pa, "main", sa)])
both, hb = timed([(pa, "encoder", sa), (pb, "main", sb)])
out["encoder+main"] = {"encoder_us": round(enc, 1), "main_us": round(main, 1), "both_us": round(both, 1), "sum_over_both": round((enc + main) / both, 3), "host_us": round(hb, 1)}
cv, _ = timed([(pa, "cv", sa)])
both, hb = timed([(pa, "encoder", sa), (pb, "cv", sb)])
out["encoder+cv"] = {"encoder_us": round(enc, 1), "cv_us": round(cv, 1), "both_us": round(both, 1), "sum_over_both": round((enc + cv) / both, 3), "host_us": round(hb, 1)}
print(json.dumps(out, indent=1))
