#!/usr/bin/env python
"""Where does wall time go beyond kernel time? Times the forward in graph / eager mode and its host phases."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from monorec_amd import MonoRecModel, synth
dev = torch.device("cuda:0")
for graph in (True, False):
    m = MonoRecModel(cv_depth_steps=32, hip_graph=graph, hip_in_flight=1); m.load_state_dict(synth.seeded_state_dict(m.state_dict())); m = m.to(dev).eval()
    b = synth.clone_batch(synth.make_batch(1, 256, 512, 2), dev)
    with torch.no_grad():
        for _ in range(5): m(dict(b))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): m(dict(b))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"graph={graph}: {1e3*(t2-t0)/30:.3f} ms/step wall, host-side enqueue {1e3*(t1-t0)/30:.3f} ms/step")
    key, plan = next(iter(m._plans.items()))
    # stage graphs alone
    if graph:
        for st in ("encoder", "encoder_tail", "cv", "main"):
            g = m._graphs[(key, st)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): g.replay()
            torch.cuda.synchronize(); print(f"  graph '{st}' replay alone: {1e3*(time.perf_counter()-t0)/20:.3f} ms")
    s = torch.cuda.current_stream()
    for st in ("encoder", "encoder_tail", "cv", "main"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): plan.run_stage(st, s.cuda_stream)
        t1 = time.perf_counter(); torch.cuda.synchronize()
        print(f"  eager '{st}': {1e3*(time.perf_counter()-t0)/10:.3f} ms (host enqueue {1e3*(t1-t0)/10:.3f})")
