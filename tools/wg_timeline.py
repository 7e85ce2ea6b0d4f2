#!/usr/bin/env python
"""Per-workgroup timeline of single conv launches (MR_CONV_DBG bit 16): every workgroup stamps the 100 MHz
clock at entry / first chunk in LDS / K loop done / stores accepted.  Shows launch ramp, DMA latency, sweep
and store tail of a layer as the chip runs it.  GPU box:  python tools/wg_timeline.py [layer,layer,...]"""
import json
import os
import sys

os.environ["MR_CONV_DBG"] = os.environ.get("MR_TL_DBG", "16")
ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT_)
from monorec_amd import build as _build  # noqa: E402
os.environ["MR_HIP_LIBRARY"] = _build.build_timeline()      # stamps are compiled in only in this diagnostic library
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorec_amd import _lib, engine, synth  # noqa: E402
from monorec_amd.model import MonoRecModel  # noqa: E402
from tools.tune_conv import build_candidate  # noqa: E402

DEFAULT = ("resnet.l1b0.conv1,resnet.l3b1.conv1,mask.enc0.0,mask.dec3.1,mask.classifier,depth.enc2.1.conv_y,"
           "depth.dec2.0,depth.enc4.1.conv_x,mask.enc3.1")


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else DEFAULT).split(",")
    m = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(m.state_dict())
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", winograd=False)
    g = torch.Generator().manual_seed(0)
    rows = {}
    for c in plan.conv_log:
        if c["name"] not in names or c["spec"] is None:
            continue
        spec = c["spec"]
        cout, cin, kh, kw = spec["w_shape"]
        nph = 1 if spec["phases"] is None else len(spec["phases"])
        srcs = [torch.randn(*s, generator=g).cuda() for s in spec["src_shapes"]]
        out = torch.empty(*spec["out_shape"], device="cuda")
        res_t = torch.randn(*spec["out_shape"], generator=g).cuda() if spec["residual"] else None
        w = torch.randn(cout, cin, kh, kw, generator=g) * 0.05
        b = torch.randn(cout, generator=g)
        pw = [torch.randn(cout, cin, kh, kw, generator=g) * 0.05 for _ in range(nph)] if nph > 1 else None
        sched = (c["mb"], c["nb"], c["split_k"], c["ck"], c.get("waves", 4), c.get("kws", 0))    # the product's schedule, split-K included
        p, fn = build_candidate(spec, sched, (srcs, out, res_t, w if nph == 1 else None, b, pw))
        desc = [k for k in p.keep if isinstance(k, _lib.ConvDesc)][0]
        wgs = p.conv_log[0]["wgs"]
        # the stamps live in the workspace, behind the split-K slabs (dbg_stamp in csrc/conv_mfma.hip)
        n_, co_, oh_, ow_ = spec["out_shape"][0], (cout + 15) // 16 * 16, spec["grid"][0], spec["grid"][1]
        slab_floats = c["split_k"] * nph * n_ * co_ * oh_ * ow_ if c["split_k"] > 1 else 0
        assert slab_floats % 2 == 0
        ws = torch.zeros(slab_floats // 2 + wgs * 12, dtype=torch.int64, device="cuda")
        stamps = ws[slab_floats // 2:]
        desc.workspace = ws.data_ptr()
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(s)
        e1.record()
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(wgs, 12).astype(np.float64)
        t = raw / 100.0     # us
        mhz = float(np.median((raw[:, 10] - raw[:, 9]) / np.maximum(t[:, 8] - t[:, 4], 1e-2)))   # clock64 ticks per us
        t0 = t[:, 0].min()
        q = lambda v: [round(float(np.percentile(v, x)), 2) for x in (0, 50, 90, 100)]
        rows[c["name"]] = dict(sched=sched, wgs=wgs, lds=c["lds"], mmac=round(c["macs"] / 1e6, 1),
                               event_us=round(e0.elapsed_time(e1) * 1e3, 1),
                               span_us=round(float(t[:, 3].max() - t0), 2),
                               start_after_first=q(t[:, 0] - t0), setup=q(t[:, 11] - t[:, 0]), first_chunk=q(t[:, 1] - t[:, 0]),
                               k_loop=q(t[:, 2] - t[:, 1]), stores=q(t[:, 3] - t[:, 2]), wg_total=q(t[:, 3] - t[:, 0]),
                               chunk_issue=q(t[:, 5] - t[:, 4]), chunk_sweep=q(t[:, 6] - t[:, 5]),
                               chunk_dma_wait=q(t[:, 7] - t[:, 6]), chunk_barrier=q(t[:, 8] - t[:, 7]),
                               clock64_ticks_per_us=round(mhz, 1))
        print(c["name"], json.dumps(rows[c["name"]]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "wg_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
