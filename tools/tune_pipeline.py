#!/usr/bin/env python
"""Schedules of the direct kernel chosen by what they do to the PIPELINED rate (four keyframes in flight), one layer at a time.

tools/tune_conv.py times every launch alone; the table it writes is the fastest schedule per layer IN ISOLATION.  With keyframes in flight a
schedule also decides how much of the chip it leaves to the other keyframes' launches (LDS footprint = workgroups per CU, workgroup count), which
an isolated timing cannot see (tools/sessions/r06_s28.sh: B8 entries measured in isolation lowered the kernel sum of the c2 shape by 4 % and
the pipelined line by 5 %).  This tool takes the runners-up of an isolated sweep (the `tried` lists of a tune_conv --report) and keeps one
where the prepare / submit / synchronize loop itself gets faster - A/B/A/B, greedy, layer by layer in order of their time:

    python tools/tune_conv.py --merge --only ... --out T.json --report R.json      (isolated sweep, writes the runners-up)
    python tools/tune_pipeline.py --report R.json --table T.json --out T2.json [--within 12] [--per-layer 3] [--margin 0.4]
"""
import argparse
import collections
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import monorec_amd  # noqa: F401,E402
import torch  # noqa: E402
from monorec_amd import MonoRecModel, engine, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--report", required=True)
    ap.add_argument("--table", default=os.path.join(ROOT, "monorec_amd", "tuned_schedules.json"))
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--within", type=float, default=12.0, help="runners-up within this many per cent of the layer's best isolated time")
    ap.add_argument("--per-layer", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.4, help="per cent the pipelined rate must gain, twice, for a runner-up to be kept")
    ap.add_argument("--budget-seconds", type=float, default=1500.0)
    a = ap.parse_args()
    dev = "cuda:0"
    table = json.load(open(a.table))
    engine.TUNED.clear()
    engine.TUNED.update({k: tuple(v) for k, v in table.items()})
    report = json.load(open(a.report))
    model = MonoRecModel(cv_depth_steps=a.depths)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).eval()
    batch = synth.clone_batch(synth.make_batch(a.batch, a.height, a.width, a.frames, seed=1), dev)
    nfl = model.hip_in_flight

    def run(n):
        pending = collections.deque()
        for _ in range(n):
            d = dict(batch)
            tok = model.prepare(d)
            if len(pending) >= nfl:
                pending.popleft().synchronize()
            pending.append(model.submit(d, tok))
        while pending:
            pending.popleft().synchronize()

    def rate():
        vals = []
        with torch.no_grad():
            run(40)
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(a.steps)
                torch.cuda.synchronize()
                vals.append(a.steps * a.batch / (time.perf_counter() - t0))
        return statistics.median(vals)

    def use(sig, sched):
        engine.TUNED[sig] = tuple(sched)
        model._invalidate()

    norm = lambda s_: (tuple(s_) + (4, 0))[:6] if len(s_) < 5 else (tuple(s_) + (0,))[:6]
    t_start = time.time()
    base = rate()
    print(f"baseline {base:.1f} keyframes/s", flush=True)
    kept, log = [], []
    for layer in sorted(report, key=lambda r: -r["us"]):
        sig = layer["sig"]
        cur = engine.TUNED.get(sig)
        if cur is None:
            continue
        best_us = layer["tried"][0][1]
        cands = [tuple(s_) for s_, us, _ in layer["tried"] if us is not None and us <= best_us * (1 + a.within / 100.0) and norm(s_) != norm(cur)][:a.per_layer]
        for cand in cands:
            if time.time() - t_start > a.budget_seconds:
                break
            use(sig, cand)
            r1 = rate()
            row = {"layer": layer["name"], "sig": sig, "table": list(cur), "candidate": list(cand), "rate": round(r1, 1), "base": round(base, 1)}
            if r1 > base * (1 + a.margin / 100.0):
                use(sig, cur)
                b2 = rate()
                use(sig, cand)
                r2 = rate()
                row.update(base_again=round(b2, 1), rate_again=round(r2, 1))
                if min(r1, r2) > max(base, b2) * (1 + 0.5 * a.margin / 100.0):
                    cur = cand
                    base = 0.5 * (r1 + r2)
                    kept.append(row)
                    row["kept"] = True
                else:
                    base = 0.5 * (base + b2)
            if engine.TUNED[sig] != tuple(cur):
                use(sig, cur)
            log.append(row)
            print(json.dumps(row), flush=True)
    out = dict(table)
    for row in kept:
        out[row["sig"]] = row["candidate"]
    with open(a.out, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(json.dumps({"kept": len(kept), "tried": len(log), "final_rate": round(rate(), 1), "seconds": round(time.time() - t_start, 1)}))


if __name__ == "__main__":
    main()
