#!/usr/bin/env python
"""CPU-only stress of the input side of an N-rank job on ONE host (`with_data_loading` at 8 ranks had no number before hardware does):

    python tools/decode_stress.py [--ranks 8] [--keyframes 200] [--pace-kfps 0] [--no-pin] [--json out.json]

Starts `--ranks` processes.  Each places itself exactly like a rank of `torchrun --nproc-per-node N bench.py` (monorec_amd.distributed.place_rank with
LOCAL_RANK / LOCAL_WORLD_SIZE: its share of the host's CPUs), opens the frame cache of the device loader with the decode-thread budget of that share
(host_thread_budget) and sweeps `--keyframes` consecutive keyframes of synthetic KITTI-sized (370 x 1226) PNGs through it - one new image per keyframe
thanks to the cache, decoded ahead on the worker threads (PIL releases the GIL), the device preprocessing replaced by a no-op.  Reported per rank: PNG
decode ms per keyframe (thread time), wall ms per keyframe and the keyframes/s the host side alone sustains - the ceiling `with_data_loading` of an
N-rank job can reach on this host - plus the aggregate.  `--pace-kfps K`: the consumer takes a keyframe every 1 / K s (a GPU that runs at K keyframes/s)
and the report says whether the read-ahead keeps up (max wait for a frame)."""
import argparse
import io
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, ranks, keyframes, pace, pin, q, go):
    os.environ.update(LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(ranks), WORLD_SIZE=str(ranks), RANK=str(rank))
    import numpy as np
    import torch
    from PIL import Image
    from monorec_amd import distributed as mrd, input_pipeline, synth
    info = mrd.place_rank(rank, ranks) if pin else {"cpus": len(os.sched_getaffinity(0)), "pinned": False}
    threads, _ = mrd.host_thread_budget(info["cpus"])
    torch.set_num_threads(1)
    pngs = []
    for i in range(8):                                   # eight distinct frames, cycled (bench.with_data_loading)
        buf = io.BytesIO()
        Image.fromarray(synth.make_u8_image(370, 1226, 3, seed=200 + i)).save(buf, format="PNG")
        pngs.append(buf.getvalue())
    decode_s = [0.0]

    def load(i):
        t = time.thread_time()
        a = np.asarray(Image.open(io.BytesIO(pngs[i % len(pngs)])))
        decode_s[0] += time.thread_time() - t
        return a
    cache = input_pipeline.FrameCache(load, lambda img: img, capacity=8, workers=threads)
    q.put(("ready", rank))
    go.wait()
    waits = []
    t0 = time.perf_counter()
    for k in range(keyframes):
        if pace > 0:
            target = t0 + k / pace
            while time.perf_counter() < target:
                time.sleep(0.0002)
        tw = time.perf_counter()
        cache.sample(k + 1, frame_count=2)
        waits.append(time.perf_counter() - tw)
    dt = time.perf_counter() - t0
    cache.close()
    q.put(("done", rank, {"rank": rank, "cpus": info["cpus"], "pinned": bool(info.get("pinned")), "decode_threads": threads,
                          "decode_ms_per_keyframe": decode_s[0] / keyframes * 1e3, "wall_ms_per_keyframe": dt / keyframes * 1e3,
                          "host_keyframes_per_s": keyframes / dt, "max_wait_ms": max(waits) * 1e3, "decoded_images_per_keyframe": (cache.decoded - 2) / keyframes}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--keyframes", type=int, default=200)
    ap.add_argument("--pace-kfps", type=float, default=0.0)
    ap.add_argument("--no-pin", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q, go = ctx.Queue(), ctx.Event()
    procs = [ctx.Process(target=worker, args=(r, a.ranks, a.keyframes, a.pace_kfps, not a.no_pin, q, go)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    for _ in procs:
        assert q.get(timeout=300)[0] == "ready"
    go.set()                                             # every rank sweeps at the same time
    rows = sorted((q.get(timeout=900)[2] for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
    out = {"host_cpus": len(os.sched_getaffinity(0)), "ranks": a.ranks, "keyframes_per_rank": a.keyframes, "pace_kfps": a.pace_kfps, "per_rank": rows,
           "aggregate_host_keyframes_per_s": sum(r["host_keyframes_per_s"] for r in rows),
           "slowest_rank_keyframes_per_s": min(r["host_keyframes_per_s"] for r in rows),
           "decode_ms_per_keyframe_mean": sum(r["decode_ms_per_keyframe"] for r in rows) / len(rows)}
    print(json.dumps(out))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
