#!/usr/bin/env python
"""Copies of the measured tables WITHOUT the entries one input shape uses (no GPU needed): how fast is a shape the tables have never seen?

    python tools/holdout_tables.py --shape B H W F D [--bf16] --out-dir gpurun_out/holdout
    MR_TUNED_SCHEDULES=<dir>/tuned_schedules.json MR_TUNED_WINOGRAD=<dir>/tuned_winograd.json MR_TUNED_B8=<dir>/tuned_b8.json python bench.py --batch B ...

The shape's launches then take the nearest-signature rules of monorec_amd.engine (nearest_schedules / nearest_form)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=5, metavar=("B", "H", "W", "F", "D"), required=True)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--out-dir", required=True)
    a = ap.parse_args()
    from monorec_amd import engine, synth
    from monorec_amd.model import MonoRecModel
    b, h, w, f, d = a.shape
    m = MonoRecModel(cv_depth_steps=d)
    plan = engine.Plan(synth.seeded_state_dict(m.state_dict()), b, h, w, f, d, (0.33, 0.0025), "cpu", bf16=int(a.bf16))
    used = {c["sig"] for c in plan.conv_log if c.get("sig")} | {c["sig"] + f"_f{int(c['f32_source'])}" for c in plan.conv_log if c.get("b8")}
    # the plan was built WITH the tables, so launches that went to a reduced-multiply form logged that form's key; their direct-kernel and
    # stride-2 keys are hidden too (every key that mentions one of the shape's output sizes at this batch size)
    sizes = {f"_o{h >> s}x{w >> s}_b{b * k}" for s in range(6) for k in (1, f)} | {f"_o{h >> s}x{w >> (s - 1)}_b{b}" for s in range(1, 6)}
    os.makedirs(a.out_dir, exist_ok=True)
    for name, table in (("tuned_schedules.json", engine.TUNED), ("tuned_winograd.json", engine.WINOGRAD), ("tuned_b8.json", engine.B8_SCHEDULES)):
        keep = {k: list(v) if isinstance(v, tuple) else v for k, v in table.items() if k not in used and not any(sz + "_" in k + "_" or k.endswith(sz) for sz in sizes)}
        json.dump(keep, open(os.path.join(a.out_dir, name), "w"), indent=0, sort_keys=True)
        print(name, len(table), "->", len(keep))


if __name__ == "__main__":
    main()
