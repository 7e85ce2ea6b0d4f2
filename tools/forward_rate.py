#!/usr/bin/env python
"""`model(data)` - what the reference's own scripts call (evaluater/evaluater.py:83, create_pointcloud.py:70) - against the in-flight-1 loop of
prepare / submit / synchronize on the same process (forward() >= 0.97 x that loop).  Interleaved rounds, median per leg.

    python tools/forward_rate.py [--rounds 5] [--steps 150]

Round 6 (tools/sessions/r06_s20.sh) used it on three host orders behind forward()'s input wait - (0) input pointers bound and the allocator bookkeeping moved so
that NOTHING stands between the wait and the encoder stage's first mr_run_launches call, the gather of the 4x4s after it; (1) round 5's order; (2) a host-side
wait for the result instead of a wait packet on the caller's stream: 0.940-0.948 / 0.938-0.945 / 0.908-0.909 x the loop.  The Python between the wait and the
first launch is not what the 0.09 ms per call are; the switch was removed again and round 5's order stays.
"""
import argparse
import collections
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import monorec_amd  # noqa: F401,E402
import torch  # noqa: E402
from monorec_amd import MonoRecModel, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=150)
a = ap.parse_args()
dev = "cuda:0"
sd = None
models = {}
for name, kw in (("forward", {}), ("forward_slot_streams", {"hip_forward_on_callers_stream": False}), ("loop1", {"hip_in_flight": 1})):
    m = MonoRecModel(cv_depth_steps=32, **kw)
    sd = sd or synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    models[name] = m.to(dev).eval()
b = synth.clone_batch(synth.make_batch(1, 256, 512, 2, seed=1), dev)


def leg_forward(m, n):
    for _ in range(n):
        out = m(dict(b))
    return out


def leg_loop1(m, n):
    pending = collections.deque()
    for _ in range(n):
        d = dict(b)
        tok = m.prepare(d)
        if pending:
            pending.popleft().synchronize()
        pending.append(m.submit(d, tok))
    return pending.popleft().synchronize()


legs = {"forward": leg_forward, "forward_slot_streams": leg_forward, "loop1": leg_loop1}
rates = {k: [] for k in legs}
with torch.no_grad():
    for k in legs:
        legs[k](models[k], 30)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k in legs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            legs[k](models[k], a.steps)
            torch.cuda.synchronize()
            rates[k].append(a.steps / (time.perf_counter() - t0))
med = {k: statistics.median(v) for k, v in rates.items()}
print(json.dumps({"forward": round(med["forward"], 1), "forward_slot_streams": round(med["forward_slot_streams"], 1), "loop1": round(med["loop1"], 1),
                  "ratio": round(med["forward"] / med["loop1"], 4), "ratio_slot_streams": round(med["forward_slot_streams"] / med["loop1"], 4), "all": {k: [round(x, 1) for x in v] for k, v in rates.items()}}))
