#!/usr/bin/env python
"""Ablation timing of single conv launches (env MR_CONV_DBG bits: 1 no sweep, 2 no input DMA, 4 no weight DMA, 8 no stores;
needs the diagnostic library: python -m monorec_amd.build --timeline, selected here through MR_HIP_LIBRARY)."""
import os, sys, json, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from monorec_amd import engine, synth
    from monorec_amd.model import MonoRecModel
    from tools.tune_conv import build_candidate, time_op
    m = MonoRecModel(cv_depth_steps=32); sd = synth.seeded_state_dict(m.state_dict())
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", winograd=False)
    g = torch.Generator().manual_seed(0)
    names = sys.argv[2].split(",")
    res = {}
    for c in plan.conv_log:
        if c["name"] not in names or c["spec"] is None: continue
        spec = c["spec"]; cout, cin, kh, kw = spec["w_shape"]
        nph = 1 if spec["phases"] is None else len(spec["phases"])
        srcs = [torch.randn(*s, generator=g).cuda() for s in spec["src_shapes"]]
        out = torch.empty(*spec["out_shape"], device="cuda")
        res_t = torch.randn(*spec["out_shape"], generator=g).cuda() if spec["residual"] else None
        w = torch.randn(cout, cin, kh, kw, generator=g) * 0.05; b = torch.randn(cout, generator=g)
        pw = [torch.randn(cout, cin, kh, kw, generator=g) * 0.05 for _ in range(nph)] if nph > 1 else None
        sched = (c["mb"], c["nb"], c["split_k"], c["ck"], c.get("waves", 4))
        p, fn = build_candidate(spec, sched, (srcs, out, res_t, w if nph == 1 else None, b, pw))
        res[c["name"]] = time_op(fn, reps=20, warm=3) * 1e6
    print("RESULT " + json.dumps(res))
else:
    names = "mask.enc0.0,mask.dec3.1,resnet.l1b0.conv1,depth.enc2.1.conv_y,resnet.l4b0.conv2,depth.dec2.0"
    table = {}
    for dbg in (0, 1, 2, 4, 8, 3, 7, 15):
        from monorec_amd import build as _build
        env = dict(os.environ, MR_CONV_DBG=str(dbg), MR_HIP_LIBRARY=_build.build_timeline())   # the switches live in the diagnostic library
        out = subprocess.run([sys.executable, __file__, "child", names], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("RESULT ")]
        table[dbg] = json.loads(line[0][7:]) if line else None
    print("dbg bits: 1 no sweep, 2 no input DMA, 4 no weight DMA, 8 no stores")
    for n in names.split(","):
        print(f"{n:22s}", "  ".join(f"{dbg}:{table[dbg][n]:6.1f}" if table[dbg] else f"{dbg}: fail" for dbg in table))
