"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product path; runs in the BUILD container, where /root/reference exists).

How much faster (or slower) is the oracle port (`oracle/monorec_oracle.py`, what bench.py's `cpu_baseline` leg times on the GPU box, kind
"port") than the UNMODIFIED reference (`MonoRecModel.forward`, /root/reference model/monorec/monorec_model.py:672-729, imported through
`oracle/ref_shims.py`) on the same host, inputs and weights?  VERDICT r4 (missing #5): the port is ~0.80 x the reference's time, so the
stated CPU baseline flatters the CPU; this script measures the ratio and writes it to `profiles/port_vs_reference.json`, and bench.py puts
`cpu_baseline.port_vs_reference` (port seconds / reference seconds) and the reference-equivalent figure on the line.  /root/reference
cannot travel to the GPU box, so the ratio is the calibration that can.

    python oracle/time_port_vs_reference.py [--threads 8] [--reps 10]

c1 = BASELINE configs[0]: 1 keyframe, 256 x 512, 2 source frames, 32 depth bins, fp32, seeded weights; 1 warm-up, then `reps` interleaved PAIRS
(reference, port, reference, port ...) at a fixed thread count so that both halves of a pair see the same machine state.  The figure quoted is the
MEDIAN of the per-pair ratios with its min - max (a single best-of ratio on a shared 8-vCPU box is +-8 %).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from monorec_amd import synth                     # noqa: E402
from oracle import monorec_oracle as orc          # noqa: E402
from oracle import ref_shims                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, torch.get_num_threads()))
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "port_vs_reference.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    Ref = ref_shims.reference_model_class()
    ref = Ref(cv_depth_steps=32).eval()
    sd = synth.seeded_state_dict(ref.state_dict(), seed=0)
    ref.load_state_dict(sd, strict=True)
    batch = synth.make_batch(1, 256, 512, 2, seed=1)

    def run_ref():
        with torch.no_grad():
            return ref(synth.clone_batch(batch))

    def run_port():
        return orc.forward(sd, batch, cv_depth_steps=32)
    out_ref, out_port = run_ref(), run_port()                 # warm-up, and the two must agree bit for bit (they are the same arithmetic)
    diff = float((out_ref["result"] - out_port["result"]).abs().max())
    assert diff == 0.0, diff
    t_ref, t_port = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter(); run_ref(); t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); run_port(); t_port.append(time.perf_counter() - t0)
    ratios = [p / r for p, r in zip(t_port, t_ref)]
    rec = {"port_seconds": statistics.median(t_port), "reference_seconds": statistics.median(t_ref), "port_vs_reference": statistics.median(ratios),
           "port_vs_reference_min": min(ratios), "port_vs_reference_max": max(ratios), "port_vs_reference_best_of": min(t_port) / min(t_ref),
           "statistic": "median of per-pair ratios (interleaved pairs, fixed thread count)",
           "threads": a.threads, "reps": a.reps, "all_reference_seconds": t_ref, "all_port_seconds": t_port,
           "workload": "c1 (BASELINE configs[0]): 1 keyframe, 256x512, 2 source frames, 32 depth bins, fp32, seeded weights",
           "host": {"cpu_count": os.cpu_count(), "torch": torch.__version__},
           "result_max_abs_diff_port_vs_reference": diff,
           "note": "port_vs_reference = oracle-port seconds / unmodified-reference seconds per keyframe on the build container's CPU (median of reps "
                   "interleaved pairs, min - max beside it); reference-equivalent keyframes/s = port keyframes/s x port_vs_reference.  The port is faster because it "
                   "skips what the reference's forward does besides arithmetic (per-call module / dict bookkeeping, the per-sample Python loop "
                   "of CostVolumeModule builds more temporaries)."}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("port_seconds", "reference_seconds", "port_vs_reference", "port_vs_reference_min", "port_vs_reference_max", "threads")}))


if __name__ == "__main__":
    main()
