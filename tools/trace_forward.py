#!/usr/bin/env python
"""A few sequential `model(data)` calls for a timeline (run under rocprofv3 --kernel-trace --memory-copy-trace):
    python tools/trace_forward.py [--in-flight 2] [--host-mats] [--steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import monorec_amd  # noqa: F401,E402  (sets GPU_MAX_HW_QUEUES before torch initialises HIP)
import torch  # noqa: E402
from monorec_amd import MonoRecModel, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--in-flight", type=int, default=2)
ap.add_argument("--host-mats", action="store_true")
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = "cuda:0"
m = MonoRecModel(cv_depth_steps=32, hip_in_flight=a.in_flight)
m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=0))
m = m.to(dev).eval()
cpu = synth.make_batch(1, 256, 512, 2, seed=1)
b = synth.clone_batch(cpu, dev)
if a.host_mats:
    for k in ("keyframe_intrinsics", "keyframe_pose", "intrinsics", "poses"):
        b[k] = cpu[k]
# where the host waits inside forward(): time in Event.synchronize, by call site
import collections
import traceback
waits = collections.defaultdict(lambda: [0, 0.0])
_orig = torch.cuda.Event.synchronize  # (the model polls with Event.query since round 3; kept for other waits)


def _sync(self):
    t = time.perf_counter()
    _orig(self)
    site = traceback.extract_stack(limit=2)[0]
    w = waits[f"{os.path.basename(site.filename)}:{site.lineno} {site.line}"]
    w[0] += 1
    w[1] += time.perf_counter() - t


torch.cuda.Event.synchronize = _sync
import monorec_amd.model as _mm  # noqa: E402
_hw = _mm._host_wait


def _hw_timed(ev):
    t = time.perf_counter()
    _hw(ev)
    site = traceback.extract_stack(limit=2)[0]
    w = waits[f"{os.path.basename(site.filename)}:{site.lineno} {site.line}"]
    w[0] += 1
    w[1] += time.perf_counter() - t


_mm._host_wait = _hw_timed
with torch.no_grad():
    for _ in range(20):
        out = m(dict(b))
    torch.cuda.synchronize()
    waits.clear()
    t0 = time.perf_counter()
    ts = []
    for _ in range(a.steps):
        t = time.perf_counter()
        out = m(dict(b))
        ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"forward(): {a.steps / dt:.1f} keyframes/s, {dt / a.steps * 1e3:.3f} ms each; host time inside forward(): median {sorted(ts)[len(ts) // 2] * 1e3:.3f} ms, max {max(ts) * 1e3:.3f} ms")
for site, (n, t) in sorted(waits.items(), key=lambda kv: -kv[1][1]):
    print(f"   {t / a.steps * 1e3:7.3f} ms/forward in {n / a.steps:.1f} waits at {site}")
print(f"   host_enqueue_stats: {m.host_enqueue_stats}")
