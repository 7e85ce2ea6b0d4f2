#!/usr/bin/env python
"""Diagnostic (GPU box): where do HIP / local-CPU oracle / build-container fixture disagree on sfcv?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import Golden
from test_gpu_kernels import _hip_cost_volume
from oracle import monorec_oracle as orc
for case in ["small", "cv_only_ragged", "d64_f4"]:
    g = Golden(case); batch = g.make_inputs()
    cv, sf = _hip_cost_volume(batch, g.depths)
    st = {}
    ocv, osf = orc.cost_volume(batch, steps=g.depths, stages=st)
    for f in range(len(sf)):
        d = (sf[f] - osf[f]).abs()
        stride = int(g.z[f"sfcv{f}.stride"]); want = torch.from_numpy(g.z[f"sfcv{f}.samples"])
        dh = (sf[f].reshape(-1)[::stride] - want).abs(); do = (osf[f].reshape(-1)[::stride] - want).abs()
        print(case, f, "hip-vs-localoracle max %.2e frac>2e-5 %.2e | hip-vs-fixture max %.2e frac %.2e | localoracle-vs-fixture max %.2e frac %.2e" % (
            d.max(), (d > 2e-5).float().mean(), dh.max(), (dh > 2e-5).float().mean(), do.max(), (do > 2e-5).float().mean()))
