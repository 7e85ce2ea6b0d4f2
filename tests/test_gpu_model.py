"""GPU (MI355X): end-to-end parity of monorec_amd.MonoRecModel against the CPU oracle and the
committed outputs of the real reference (tests/golden), through the drop-in dict API."""
import os

import pytest
import torch

from golden_util import Golden
from monorec_amd import engine, synth
from monorec_amd.model import MonoRecModel
from oracle import monorec_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RESULT_ATOL = 1e-4          # BASELINE.json north_star: depths within 1e-4 abs of the reference CPU output
BF16_MAX_ERR, BF16_MEAN_ERR = 4e-3, 6e-4   # accuracy bar of the bf16 MFMA mode (hip_bf16=True, BASELINE configs[4]): 2 x what the MI355X measures (2e-3 max /
                                           # 3e-4 mean against the fp32 CPU output; round 4's self-declared 1e-2 / 1e-3 would have let a 5 x regression pass)


def _model(depths, graph, in_flight=1, family="he", **kw):
    m = MonoRecModel(cv_depth_steps=depths, hip_graph=graph, hip_in_flight=in_flight, **kw)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0, family=family)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def _to_dev(batch):
    return synth.clone_batch(batch, DEV)


def _check_against(out, ref_out, tag):
    errs = {}
    errs["result"] = (out["result"].cpu() - ref_out["result"]).abs().max().item()
    errs["cv_mask"] = (out["cv_mask"].cpu() - ref_out["cv_mask"]).abs().max().item()
    for i in range(4):
        errs[f"pred{i}"] = (out["predicted_inverse_depths"][i].cpu() - ref_out["predicted_inverse_depths"][i]).abs().max().item()
    for i in range(5):
        a, b = out["image_features"][i].cpu(), ref_out["image_features"][i]
        errs[f"feat{i}"] = ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()
    cvd = (out["cost_volume"].cpu() - ref_out["cost_volume"]).abs()
    errs["cv_outliers"] = (cvd > 2e-4).float().mean().item()
    for f, s in enumerate(out["single_frame_cvs"]):
        # local-CPU oracle noise on another host CPU is ~4e-5 on 0.1 % of entries (see test_gpu_kernels.py)
        errs[f"sfcv{f}_outliers"] = ((s.cpu() - ref_out["single_frame_cvs"][f]).abs() > 1e-4).float().mean().item()
    print(tag, {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["result"] <= RESULT_ATOL, errs
    assert errs["cv_mask"] <= 1e-4, errs
    assert all(errs[f"pred{i}"] <= RESULT_ATOL for i in range(4)), errs
    assert all(errs[f"feat{i}"] <= 1e-4 for i in range(5)), errs
    assert errs["cv_outliers"] <= 5e-4, errs
    assert all(errs[f"sfcv{f}_outliers"] <= 1e-4 for f in range(len(out["single_frame_cvs"]))), errs
    return errs


@pytest.mark.parametrize("case", ["small", "small_hard_pose", "d64_f4"])
def test_forward_matches_oracle_and_fixture(hip_lib, case):
    g = Golden(case)
    model, sd = _model(g.depths, graph=False)
    batch = g.make_inputs()
    with torch.no_grad():
        out = model(_to_dev(batch))
    torch.cuda.synchronize()
    ref_out = orc.forward(sd, batch, cv_depth_steps=g.depths)
    _check_against(out, ref_out, case)
    # committed outputs of the real reference
    g.compare("result", out["result"], atol=RESULT_ATOL)
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    for i in range(4):
        g.compare(f"pred{i}", out["predicted_inverse_depths"][i], atol=RESULT_ATOL)
    for i in range(5):
        g.compare(f"feat{i}", out["image_features"][i], atol=2e-4, rtol=1e-4)
    # hard pose: host-CPU dependent LAPACK rounding of inverse(pose) is amplified by the reference's own fp32
    # cancellation (see test_gpu_kernels.py); everything downstream of the cost volume is unaffected at 1e-4
    cv_atol, sf_atol = (2e-4, 1e-4) if g.hard_pose else (1e-5, 2e-6)
    g.compare("cost_volume", out["cost_volume"], atol=cv_atol, max_outlier_frac=1e-4)
    for f in range(g.frames):
        g.compare(f"sfcv{f}", out["single_frame_cvs"][f], atol=sf_atol, max_outlier_frac=1e-4)
    # dict contract (monorec_model.py:675-677,726-727)
    assert out["result"] is out["predicted_inverse_depths"][0] and out["mask"] is out["cv_mask"]
    assert out["cv_depth_steps"].dtype == torch.int32 and int(out["cv_depth_steps"][0]) == g.depths
    assert abs(float(out["inv_depth_min"][0]) - 0.33) < 1e-6 and "cv_module_time" in out
    assert [tuple(t.shape[1:]) for t in out["image_features"]] == [(64, g.h // 2, g.w // 2), (64, g.h // 4, g.w // 4),
                                                                   (128, g.h // 8, g.w // 8), (256, g.h // 16, g.w // 16),
                                                                   (512, g.h // 32, g.w // 32)]


def test_c2_config_matches_reference_fixture_and_graph_replay(hip_lib):
    """BASELINE configs[1]: single keyframe 256x512, 2 source frames, 32 bins, fp32, <= 1e-4 vs CPU."""
    g = Golden("c1_256x512")
    batch = g.make_inputs()
    eager, sd = _model(g.depths, graph=False)
    with torch.no_grad():
        out_e = eager(_to_dev(batch))
    res_e = out_e["result"].clone()
    info = g.compare("result", res_e, atol=RESULT_ATOL)
    print("c2 result vs reference fixture", info)
    g.compare("cv_mask", out_e["cv_mask"], atol=1e-4)
    g.compare("cost_volume", out_e["cost_volume"], atol=1e-4, max_outlier_frac=5e-4)
    for i in range(5):
        g.compare(f"feat{i}", out_e["image_features"][i], atol=2e-4, rtol=1e-4)
    # hipGraph replay (3 calls: eager warm-up, capture, replay) must reproduce the eager result bit for bit
    graphed, _ = _model(g.depths, graph=True)
    with torch.no_grad():
        for _ in range(4):
            out_g = graphed(_to_dev(batch))
    torch.cuda.synchronize()
    assert torch.equal(out_g["result"], res_e)
    # a different keyframe through the captured graphs still matches a fresh eager run
    batch2 = synth.make_batch(1, 256, 512, 2, seed=9)
    with torch.no_grad():
        r_g = graphed(_to_dev(batch2))["result"].clone()
        r_e = eager(_to_dev(batch2))["result"].clone()
    assert torch.equal(r_g, r_e)


def test_c2_config_with_separable_cost_volume_sums(hip_lib):
    """MonoRecModel(hip_cv_separable=True): the fp32 path's opt-in - the cost volume through mr_cost_volume_relaxed_f32 (separable
    3x3 window sums), everything else unchanged.  Against the committed output of the reference at BASELINE configs[1]: the depth stays inside
    north_star's 1e-4 bar (measured ~2e-6), the mask at its bar; the volumes move by <= 2e-4 (exact-order default: 5e-7) with the validity - the
    zeros - exactly that of the default plan."""
    g = Golden("c1_256x512")
    batch = g.make_inputs()
    sep, sd = _model(g.depths, graph=False, hip_cv_separable=True)
    ref_plan, _ = _model(g.depths, graph=False)
    with torch.no_grad():
        out = sep(_to_dev(batch))
        base = ref_plan(_to_dev(batch))
    torch.cuda.synchronize()
    info = g.compare("result", out["result"], atol=RESULT_ATOL)
    print("c2, separable cost-volume sums: result vs reference fixture", info,
          "vs the default plan %.2e" % (out["result"] - base["result"]).abs().max().item())
    assert info["max_abs"] <= 5e-6                     # measured 1.4e-6 (the default plan: 1.3e-6): far inside the 1e-4 bar, and pinned well below it
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    for f in range(g.frames):
        g.compare(f"sfcv{f}", out["single_frame_cvs"][f], atol=1e-4)
        assert torch.equal((out["single_frame_cvs"][f] == 0).all(1), (base["single_frame_cvs"][f] == 0).all(1))   # not one validity flip (all-depth zero = invalid)
        assert float((out["single_frame_cvs"][f] - base["single_frame_cvs"][f]).abs().max()) > 0    # (the relaxed kernel did run)
    g.compare("cost_volume", out["cost_volume"], atol=2e-4, max_outlier_frac=1e-3)


def test_reference_example_sample_with_metrics(hip_lib):
    """The reference's own example (KITTI 07 / image 169, DVSO poses, lidar ground truth).

    Real poses sit ~80 m from the origin, so `inverse(pose_f) @ pose_kf` cancels in fp32 and the warp lands a few
    1e-5 px differently on every host CPU (LAPACK/MKL kernels differ); a handful of pixels then flip the all-depth
    validity mask, and the (un-normalised, randomly initialised) depth net spreads each flip over its receptive
    field.  Measured on the MI355X box: the *same CPU oracle* there deviates from the build container's reference
    output by up to 6e-3 in `result` (2 % of pixels > 1e-4).  End-to-end equality at 1e-4 is therefore not defined
    for this sample even CPU-vs-CPU; parity is asserted stage by stage, each stage fed with identical inputs."""
    from monorec_amd import metrics
    g = Golden("kitti_example_169")
    batch = g.make_inputs()
    model, sd = _model(g.depths, graph=False)
    with torch.no_grad():
        out = model(_to_dev(batch))
    torch.cuda.synchronize()
    # 1. pose-independent stage: bit-level noise only
    for i in range(5):
        g.compare(f"feat{i}", out["image_features"][i], atol=2e-4, rtol=1e-4)
    # 2. cost volume vs the committed reference output: equal up to validity flips
    for f in range(g.frames):
        g.compare(f"sfcv{f}", out["single_frame_cvs"][f], atol=1e-4, max_outlier_frac=5e-4)
    g.compare("cost_volume", out["cost_volume"], atol=2e-4, max_outlier_frac=1e-3)
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    # 3. everything downstream of the cost volume, oracle fed with this run's cost volume / features: 1e-4
    feats = [t.cpu() for t in out["image_features"]]
    sfcvs = [t.cpu() for t in out["single_frame_cvs"]]
    mask_ref = orc.mask_module(sd, sfcvs, feats)
    assert (out["cv_mask"].cpu() - mask_ref).abs().max().item() <= 1e-4
    preds = orc.depth_module(sd, out["cost_volume"].cpu(), batch["keyframe"], feats)
    lo, hi = 0.0025, 0.33
    for i in range(4):
        want = (1 - preds[i]) * lo + preds[i] * hi
        err = (out["predicted_inverse_depths"][i].cpu() - want).abs().max().item()
        assert err <= RESULT_ATOL, (i, err)
    # 4. end to end vs the committed reference output: bounded by the reference's own CPU-to-CPU spread (see above)
    d = (out["result"].cpu() - torch.from_numpy(g.z["result.full"])).abs()
    print("kitti example: result vs reference fixture max %.2e, p99 %.2e, frac>1e-4 %.3f" %
          (d.max(), d.flatten().kthvalue(int(0.99 * d.numel())).values, (d > 1e-4).float().mean()))
    assert d.max().item() <= 5e-2 and (d > 1e-3).float().mean().item() <= 0.02
    # 5. the seven sparse metrics on device vs the oracle on the same prediction (tight) and vs the reference's values
    data = {"result": out["result"], "target": g.target().to(DEV)}
    want_here = orc.sparse_metrics(out["result"].cpu(), g.target(), None, 80)
    for fn, want in g.reference_metrics().items():
        got = float(getattr(metrics, fn)(data, None, 80))
        assert abs(got - float(want_here[fn])) <= 2e-5 * max(1.0, abs(want)), (fn, got, float(want_here[fn]))
        assert abs(got - want) <= 1e-2 * max(1.0, abs(want)), (fn, got, want)


def test_reference_example_sample_with_the_fixtures_own_matrices(hip_lib):
    """The real sample again, with the one host-dependent step taken out: the fixture stores the 3x3 / 3x4 matrices the reference
    formed on the generating host (monorec_model.py:171,198,207; `geom.kinv`, `geom.proj`) and its full all-depth validity maps
    (`geom.valid_bits`); the matrices are handed to the cost-volume launch instead of this host's LAPACK result.  Then, on real data:
      1. single-frame volumes within 2e-6 of the reference, validity identical on every pixel of both frames;
      2. the fused volume equals the reference's fusion formula (:257-269) evaluated on these single-frame volumes, except on the
         few pixels whose frame weights vanish to within rounding: there `sum(w) != 0` (:265) is decided by the last bit of
         `sum_d exp(..)` - MKL's vmsExp in the reference, ocml's expf here - and the reference itself jumps between 0 and
         1 - 2 sad (measured: 10 sky pixels, all with sum(w) <= 1.2e-7 = one ulp of 1);
      3. mask, image features and - given this run's fused volume - every depth scale within 1e-4 (stage tests, here and in
         test_reference_example_sample_with_metrics);
      4. end to end the depth therefore equals the reference's outside the footprint of those pixels; the rest is reported."""
    import numpy as np
    g = Golden("kitti_example_169")
    batch = g.make_inputs()
    model, sd = _model(g.depths, graph=False)
    model._geometry_override = (torch.from_numpy(g.z["geom.kinv"]), torch.from_numpy(g.z["geom.proj"]))
    with torch.no_grad():
        out = model.submit(_to_dev(batch)).synchronize()      # zero-copy interface: step 5 patches the plan's resident volume and re-runs its launches
    torch.cuda.synchronize()
    sf = [t.cpu() for t in out["single_frame_cvs"]]
    nf, h, w = g.frames, g.h, g.w
    # 1. single-frame volumes and validity
    for f in range(nf):
        info = g.compare(f"sfcv{f}", sf[f], atol=2e-6)
        print("kitti example, fixture matrices: sfcv%d vs reference" % f, info)
    ref_valid = np.unpackbits(g.z["geom.valid_bits"])[: nf * h * w].reshape(nf, h, w).astype(bool)
    hip_valid = np.stack([~(t[0] == 0).all(0).numpy() for t in sf])
    assert np.array_equal(hip_valid, ref_valid), f"{int((hip_valid != ref_valid).sum())} validity flips"
    # 2. fusion formula of the reference on these volumes (sad = (1 - sfcv) / 2 where valid)
    valid_t = torch.from_numpy(hip_valid).unsqueeze(1).float()
    sad = torch.stack([(1 - t[0]) / 2 for t in sf])
    e = torch.exp(-10 * torch.pow(sad - sad.min(1, keepdim=True)[0], 2))
    wgt = (1 - 1 / (g.depths - 1) * (e.sum(1, keepdim=True) - 1)) * valid_t
    fused = (sad * wgt).sum(0)
    wsum = wgt.sum(0).squeeze()
    nz = wsum != 0
    fused[:, nz] /= wsum[nz]
    fused = 1 - 2 * fused
    fused[:, ~nz] = 0
    mask = out["cv_mask"].cpu()
    dev = (out["cost_volume"].cpu()[0] - fused * (1 - mask[0])).abs().amax(0)
    off = dev > 1e-4
    print("fused volume vs the reference formula on these single-frame volumes: %d pixels differ, sum(w) there <= %.2e; elsewhere max %.1e"
          % (int(off.sum()), float(wsum[off].abs().max()) if off.any() else 0.0, float(dev[~off].max())))
    assert int(off.sum()) <= 64 and (not off.any() or float(wsum[off].abs().max()) <= 1e-6)
    info = g.compare("cost_volume", out["cost_volume"], atol=1e-5, max_outlier_frac=2e-4)
    print("masked fused volume vs reference fixture samples:", info)
    # 3. mask / features vs the fixture, depth scales vs the oracle's depth module fed with this run's volume
    m = (mask - torch.from_numpy(g.z["cv_mask.full"])).abs()
    assert m.max().item() <= 1e-4
    for i in range(5):
        g.compare(f"feat{i}", out["image_features"][i], atol=1e-5, rtol=1e-5)
    preds = orc.depth_module(sd, out["cost_volume"].cpu(), batch["keyframe"], [t.cpu() for t in out["image_features"]])
    for i in range(4):
        want = (1 - preds[i]) * 0.0025 + preds[i] * 0.33
        assert (out["predicted_inverse_depths"][i].cpu() - want).abs().max().item() <= RESULT_ATOL, i
    # 4. end to end, for the record
    d = (out["result"].cpu() - torch.from_numpy(g.z["result.full"])).abs()
    print("kitti example, fixture matrices: result vs reference max %.2e, frac > 1e-4 %.3f (footprint of the undetermined pixels), cv_mask max %.2e"
          % (d.max(), (d > 1e-4).float().mean(), m.max()))
    assert d.max().item() <= 5e-2 and (d > 1e-4).float().mean().item() <= 0.15
    # 5. the composed statement: the deviating pixels are among those the fixture lists as undetermined (the
    #    reference's own sum(w) <= 1e-6 there); with the reference's values written over exactly these pixels of the masked
    #    volume (:713) and the depth module run again on it, the end-to-end depth of the real sample is inside the 1e-4 bar
    low_idx = torch.from_numpy(g.z["fuse.lowweight_idx"]).long()
    off_idx = torch.nonzero(off.reshape(-1)).reshape(-1)
    assert set(off_idx.tolist()) <= set(low_idx.tolist()), "a pixel outside the undetermined set deviates"
    where = torch.searchsorted(low_idx, off_idx)
    patch = torch.from_numpy(g.z["fuse.lowweight_cv"])[where]                       # (n_off, D): the reference's masked volume there
    plan = next(iter(model._plans.values()))
    cvbuf = plan.buf["cost_volume"]
    if off_idx.numel():
        cvbuf[0].view(g.depths, -1)[:, off_idx.to(DEV)] = patch.t().contiguous().to(DEV)
    names = [n for n, _ in plan.stages["main"]]
    first = names.index("mask.classifier") + 1
    stream = torch.cuda.current_stream()
    for _, fn in plan.stages["main"][first:]:
        fn(stream.cuda_stream)
    torch.cuda.synchronize()
    d2 = (plan.preds[0].cpu() - torch.from_numpy(g.z["result.full"])).abs()
    print("kitti example, fixture matrices, %d undetermined pixels patched: result vs reference max %.2e" % (off_idx.numel(), d2.max()))
    assert d2.max().item() <= RESULT_ATOL


def test_c3_full_shape_against_the_oracle(hip_lib):
    """BASELINE configs[2] at its full size: batch 8, 256x512, 4 source frames, 64 depth bins (the shape the c3 bench line and
    its rocprof counters are quoted on) - HIP path vs the CPU oracle on this host, `result` <= 1e-4."""
    model, sd = _model(64, graph=False)
    batch = synth.make_batch(8, 256, 512, 4, seed=3)
    with torch.no_grad():
        out = model(_to_dev(batch))
        out = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in out.items() if k in
               ("result", "cv_mask", "predicted_inverse_depths", "image_features", "cost_volume", "single_frame_cvs")}
    torch.cuda.synchronize()
    ref_out = orc.forward(sd, batch, cv_depth_steps=64)
    assert out["result"].shape == (8, 1, 256, 512) and len(out["single_frame_cvs"]) == 4
    _check_against(out, ref_out, "c3 full shape")


@pytest.mark.parametrize("batch_size", [4, 1])
def test_oxford_robotcar_evaluation_shape_against_the_oracle(hip_lib, batch_size):
    """The reference's second evaluation config (configs/evaluate/eval_monorec_oxrc.json:26: batch_size 4, frame_count 2; the loader feeds 320x640,
    data_loader/oxford_robotcar_dataset.py:53) on its measured tables (tools/sessions/r05_s18.sh: every kernel family appears, the split F(4x4,3x3)
    kernel included) against the CPU oracle at the 1e-4 bar; batch 1 = what nn.DataParallel / one request at a time sees."""
    model, sd = _model(32, graph=False)
    batch = synth.make_batch(batch_size, 320, 640, 2, seed=6)
    with torch.no_grad():
        out = model(_to_dev(batch))
        out = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in out.items() if k in
               ("result", "cv_mask", "predicted_inverse_depths", "image_features", "cost_volume", "single_frame_cvs")}
    torch.cuda.synchronize()
    plan = next(iter(model._plans.values()))
    assert any(c.get("wino_variant") == 4 for c in plan.conv_log) or batch_size == 1          # the tables of this shape select the split kernel (batch 4)
    assert sum(1 for c in plan.conv_log if c.get("stride2")) >= 2
    ref_out = orc.forward(sd, batch, cv_depth_steps=32)
    assert out["result"].shape == (batch_size, 1, 320, 640) and len(out["single_frame_cvs"]) == 2
    _check_against(out, ref_out, "320x640 batch %d" % batch_size)


@pytest.mark.parametrize("shape", [(480, 640), (192, 448)], ids=["tum_480x640_tabled", "192x448_no_entries"])
def test_other_input_shapes_against_the_oracle(hip_lib, shape):
    """Shapes beyond the KITTI / Oxford crops.  480x640 (TUM RGB-D class, data_loader/tum_rgbd_dataset.py) runs on the entries
    tools/tune_all.py measured for it (tools/sessions/r06_s23.sh); 192x448 has no entry in any table: every launch is chosen by the
    nearest-signature rules (engine.nearest_schedules / nearest_form) and validated against its own geometry by the library.  Both at the
    1e-4 bar of the path."""
    h, w = shape
    model, sd = _model(32, graph=False)
    batch = synth.make_batch(1, h, w, 2, seed=9)
    with torch.no_grad():
        out = model(_to_dev(batch))
        out = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in out.items() if k in
               ("result", "cv_mask", "predicted_inverse_depths", "image_features", "cost_volume", "single_frame_cvs")}
    torch.cuda.synchronize()
    plan = next(p for k, p in model._plans.items() if k[2] == h and k[3] == w)
    tabled = [c["sig"] for c in plan.conv_log if c.get("sig") in engine.TUNED]
    if shape == (192, 448):
        assert not tabled, tabled[:3]
        assert any(c.get("winograd") for c in plan.conv_log)               # the rules still send layers to the reduced-multiply forms
    else:
        assert len(tabled) >= 20
    ref_out = orc.forward(sd, batch, cv_depth_steps=32)
    assert out["result"].shape == (1, 1, h, w)
    _check_against(out, ref_out, "%dx%d" % shape)


def test_c5_shape_in_fp32_and_bf16(hip_lib):
    """BASELINE configs[4] shape (512x1024, 4 source frames, 48 depth bins) against the CPU oracle: the fp32 path at the 1e-4
    bar, and the bf16 MFMA mode the configuration names (hip_bf16=True: bf16 operands, fp32 accumulate, fp32 storage) at its own
    stated bar - depth within 1e-2 everywhere and 1e-3 on average."""
    model, sd = _model(48, graph=False)
    batch = synth.make_batch(1, 512, 1024, 4, seed=5)
    with torch.no_grad():
        out = model(_to_dev(batch))
    torch.cuda.synchronize()
    ref_out = orc.forward(sd, batch, cv_depth_steps=48)
    assert out["result"].shape == (1, 1, 512, 1024) and len(out["single_frame_cvs"]) == 4
    _check_against(out, ref_out, "c5-shape fp32")
    del model, out
    m16 = MonoRecModel(cv_depth_steps=48, hip_in_flight=1, hip_bf16=True)
    m16.load_state_dict(sd)
    m16 = m16.to(DEV).eval()
    with torch.no_grad():
        o16 = m16(_to_dev(batch))
    torch.cuda.synchronize()
    err = (o16["result"].cpu() - ref_out["result"]).abs()
    print("c5 shape, bf16 MFMA mode: result max|err| %.3e mean %.3e" % (err.max(), err.mean()))
    assert err.max().item() <= BF16_MAX_ERR and err.mean().item() <= BF16_MEAN_ERR


def test_use_stereo_adds_a_source_view(hip_lib):
    """use_stereo=True (monorec_model.py:164-167): the stereo frame joins the source views; against the committed output of
    the reference built with use_stereo=True and against the oracle fed with three source frames."""
    g = Golden("small_stereo")
    b3 = g.make_inputs()
    m = MonoRecModel(cv_depth_steps=g.depths, use_stereo=True, hip_in_flight=1)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    data = _to_dev(b3)
    data["stereoframe"], data["stereoframe_intrinsics"], data["stereoframe_pose"] = \
        data["frames"].pop(), data["intrinsics"].pop(), data["poses"].pop()
    with torch.no_grad():
        out = m(data)
    torch.cuda.synchronize()
    assert len(out["single_frame_cvs"]) == 3
    g.compare("result", out["result"], atol=RESULT_ATOL)
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    g.compare("cost_volume", out["cost_volume"], atol=1e-5, max_outlier_frac=1e-4)
    _check_against(out, orc.forward(sd, b3, cv_depth_steps=g.depths), "use_stereo")
    # stereo only (use_mono=False): a single source view
    m1 = MonoRecModel(cv_depth_steps=g.depths, use_stereo=True, use_mono=False, hip_in_flight=1)
    m1.load_state_dict(sd)
    m1 = m1.to(DEV).eval()
    with torch.no_grad():
        out1 = m1(dict(data))
    one = {k: v for k, v in b3.items()}
    one["frames"], one["intrinsics"], one["poses"] = b3["frames"][2:], b3["intrinsics"][2:], b3["poses"][2:]
    assert len(out1["single_frame_cvs"]) == 1
    _check_against(out1, orc.forward(sd, one, cv_depth_steps=g.depths), "stereo only")


def test_cv_depths_through_the_model(hip_lib):
    """The optional `cv_depths` entry of the input dict reaches the cost-volume kernel (and is not sticky)."""
    model, sd = _model(8, graph=False)
    batch = synth.make_batch(1, 64, 96, 2, seed=4)
    pix = synth.make_pixel_depths(1, 8, 64, 96, seed=35)
    data = _to_dev(batch)
    data["cv_depths"] = pix.to(DEV)
    with torch.no_grad():
        out = model(data)
    torch.cuda.synchronize()
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=8, cv_depths=pix), "cv_depths")     # before the buffers are reused
    with torch.no_grad():
        plain = model(_to_dev(batch))["result"].clone()
    ref_plain = orc.forward(sd, batch, cv_depth_steps=8)
    assert (plain.cpu() - ref_plain["result"]).abs().max().item() <= RESULT_ATOL


def test_sfcv_without_mult_mask_through_the_model(hip_lib):
    """sfcv_mult_mask=False (monorec_model.py:252-253) reaches the kernel; the MaskModule consumes the differently masked volumes."""
    batch = synth.make_batch(1, 64, 96, 2, seed=37, hard_pose=False)
    m = MonoRecModel(cv_depth_steps=8, sfcv_mult_mask=False, hip_in_flight=1)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=8, sfcv_mult_mask=False), "sfcv_mult_mask=False")


OPTION_CASES = {"pm1": dict(pretrain_mode=1), "pm2": dict(pretrain_mode=2), "pm3": dict(pretrain_mode=3),
                "nocv": dict(no_cv=True), "mask_nocv": dict(mask_use_cv=False), "mask_nofeats": dict(mask_use_feats=False),
                "simple": dict(simple_mask=True)}


@pytest.mark.parametrize("case", sorted(OPTION_CASES))
def test_model_options_against_reference_fixture(hip_lib, case):
    """Eval-mode branches of monorec_model.py:680-727 (pretrain_mode 1/2/3, no_cv) and :352-355 (mask_use_cv/feats=False):
    plan-level variants of the same kernels, checked against outputs of the reference model (fixture small_options)."""
    g = Golden("small_options")
    kw = OPTION_CASES[case]
    batch = g.make_inputs()
    batch["mvobj_mask"] = torch.from_numpy(g.z["input.mvobj_mask"])
    if case == "simple":                      # SimpleMaskModule reads a previous prediction from the dict (monorec_model.py:453)
        batch["predicted_inverse_depths"] = [torch.from_numpy(g.z["input.prev_depth"])]
    m = MonoRecModel(cv_depth_steps=g.depths, hip_in_flight=1, **kw)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    data = _to_dev(batch)
    data["mvobj_mask"] = batch["mvobj_mask"].to(DEV)
    if case == "simple":
        data["predicted_inverse_depths"] = [batch["predicted_inverse_depths"][0].to(DEV)]
    with torch.no_grad():
        out = m(data)
    torch.cuda.synchronize()
    g.compare(f"{case}.result", out["result"], atol=RESULT_ATOL)
    g.compare(f"{case}.cv_mask", out["cv_mask"], atol=1e-4)
    g.compare(f"{case}.cost_volume", out["cost_volume"], atol=2e-4, max_outlier_frac=5e-4)
    for f in range(len(out["single_frame_cvs"])):
        g.compare(f"{case}.sfcv{f}", out["single_frame_cvs"][f], atol=1e-4, max_outlier_frac=1e-4)
    assert ("predicted_inverse_depths" in out) == (case != "pm2") and ("mask" in out) == (case != "pm2")
    ref = orc.forward(sd, batch, cv_depth_steps=g.depths, **kw)
    assert (out["result"].cpu() - ref["result"]).abs().max().item() <= RESULT_ATOL
    if case != "pm2":
        for i in range(4):
            assert (out["predicted_inverse_depths"][i].cpu() - ref["predicted_inverse_depths"][i]).abs().max().item() <= RESULT_ATOL
    if case == "nocv":
        assert float(out["cost_volume"].abs().max()) == 0.0 and len(out["single_frame_cvs"]) == len(batch["poses"])


def test_cv_patch_size_through_the_model(hip_lib):
    batch = synth.make_batch(1, 64, 96, 2, seed=45, hard_pose=False)
    m = MonoRecModel(cv_depth_steps=8, cv_patch_size=5, hip_in_flight=1)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    ref = orc.forward(sd, batch, cv_depth_steps=8, cv_patch_size=5)
    _check_against(out, ref, "cv_patch_size=5")


@pytest.mark.parametrize("depths", [6, 7, 13])
def test_any_number_of_depth_hypotheses_through_the_model(hip_lib, depths):
    """cv_depth_steps that are not multiples of 4, odd ones included (the reference takes any: monorec_model.py:184): the whole forward
    against the oracle (rounds 1-3 raised NotImplementedError)."""
    batch = synth.make_batch(1, 64, 96, 2, seed=21)
    m, sd = _model(depths, False)
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    assert out["cost_volume"].shape[1] == depths and all(s.shape[1] == depths for s in out["single_frame_cvs"])
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=depths), f"D={depths}")


def test_depth_large_model(hip_lib):
    """depth_large_model=True (monorec_model.py:482-483): the plan takes the DepthModule widths from the weights."""
    g = Golden("small_large_depth")
    batch = g.make_inputs()
    m = MonoRecModel(cv_depth_steps=g.depths, depth_large_model=True, hip_in_flight=1)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    g.compare("result", out["result"], atol=RESULT_ATOL)
    for i in range(4):
        g.compare(f"pred{i}", out["predicted_inverse_depths"][i], atol=RESULT_ATOL)
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=g.depths), "depth_large_model")


def test_bf16_mode_end_to_end(hip_lib):
    """hip_bf16=True (BASELINE configs[4] numerics): convolutions on the bf16 MFMA, cost volume and storage fp32.  Not within the
    1e-4 bar by construction; the test pins how far off it is and that the fp32 default is untouched."""
    m = MonoRecModel(cv_depth_steps=16, hip_in_flight=1, hip_bf16=True)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    batch = synth.make_batch(1, 128, 192, 2, seed=3)
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    ref = orc.forward(sd, batch, cv_depth_steps=16)
    err = (out["result"].cpu() - ref["result"]).abs()
    cverr = (out["cost_volume"].cpu() - ref["cost_volume"]).abs().max().item()
    print("bf16 mode: result max|err| %.3e mean %.3e; cv_mask max %.3e" %
          (err.max(), err.mean(), (out["cv_mask"].cpu() - ref["cv_mask"]).abs().max()))
    assert torch.isfinite(out["result"]).all()
    # the stated accuracy bar of the bf16 MFMA mode (BASELINE configs[4] numerics: bf16 operands, fp32 accumulate): depth within
    # 1e-2 of the fp32 CPU output everywhere and 1e-3 on average (measured: 2e-3 / 3e-4) - bf16-sized, not garbage, not fp32
    assert 1e-6 < err.max().item() <= BF16_MAX_ERR and err.mean().item() <= BF16_MEAN_ERR
    # the unmasked part of the cost volume comes from the fp32 cost-volume kernel; only the bf16 mask scales it
    assert cverr < 5e-2


@pytest.mark.parametrize("depths", [7, 20])
def test_bf16_mode_with_depth_counts_off_the_fast_paths(hip_lib, depths):
    """hip_bf16=True with a depth count that has no register-resident fusion kernel (the B8 copies of the volumes then come from the
    layout-conversion kernels, channel counts that are not multiples of 8 included): same accuracy bar as the 16-hypotheses case."""
    m = MonoRecModel(cv_depth_steps=depths, hip_in_flight=1, hip_bf16=True)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    batch = synth.make_batch(1, 64, 96, 2, seed=5)
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    ref = orc.forward(sd, batch, cv_depth_steps=depths)
    err = (out["result"].cpu() - ref["result"]).abs()
    assert torch.isfinite(out["result"]).all() and out["cost_volume"].shape[1] == depths
    assert err.max().item() <= BF16_MAX_ERR and err.mean().item() <= BF16_MEAN_ERR, (err.max().item(), err.mean().item())
    assert (out["cost_volume"].cpu() - ref["cost_volume"]).abs().max().item() < 5e-2


def test_bf16x3_mode_end_to_end(hip_lib):
    """hip_bf16x3=True: convolutions as three bf16 MFMAs over hi/lo splits.  CPU emulation of this arithmetic over the whole network
    gives 4e-6 on the depth (fp32 path: 1.3e-6), the MI355X 3.7e-6: it has to meet the same 1e-4 bar as the fp32 default.
    (A secondary mode: the reference multiplies fp32 by fp32, so the fp32 MFMA path stays the default and the headline.)"""
    m = MonoRecModel(cv_depth_steps=16, hip_in_flight=1, hip_bf16x3=True)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    batch = synth.make_batch(1, 128, 192, 2, seed=3)
    with torch.no_grad():
        out = m(_to_dev(batch))
    torch.cuda.synchronize()
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=16), "bf16x3")


def test_batch_independence(hip_lib):
    """Keyframes are independent: sample i of a batch equals the same sample run alone."""
    model, _ = _model(8, graph=False)
    batch = synth.make_batch(3, 64, 96, 2, seed=21)
    with torch.no_grad():
        full = model(_to_dev(batch))["result"].clone()
        for i in (0, 2):
            one = {k: (v[i:i + 1] if torch.is_tensor(v) else [t[i:i + 1] for t in v]) for k, v in batch.items()}
            alone = model(_to_dev(one))["result"]
            # the launch schedule (split-K) depends on the batch size, so allow summation-order noise
            assert (alone[0] - full[i]).abs().max().item() < 1e-6


def test_skip_dead_layer4_leaves_every_other_output_bit_identical(hip_lib):
    """MonoRecModel(hip_skip_dead_layer4=True): ResNet layer4 (monorec_model.py:118-129) feeds only image_features[4], which neither
    MaskModule (:372-380) nor DepthModule (:545) reads - not launching it must not move a bit of anything else; forward() and submit() both."""
    full, _ = _model(8, graph=False)
    lean, _ = _model(8, graph=False, hip_skip_dead_layer4=True)
    batch = synth.make_batch(2, 64, 96, 2, seed=33)
    with torch.no_grad():
        a = full(_to_dev(batch))
        b = lean(_to_dev(batch))
        c = lean.submit(_to_dev(batch)).result()
    torch.cuda.synchronize()
    assert len(a["image_features"]) == 5 and len(b["image_features"]) == 4 and len(c["image_features"]) == 4
    for out in (b, c):
        for k in ("result", "cv_mask", "cost_volume"):
            assert torch.equal(out[k], a[k]), k
        for i in range(4):
            assert torch.equal(out["image_features"][i], a["image_features"][i]) and torch.equal(out["predicted_inverse_depths"][i], a["predicted_inverse_depths"][i])
        assert all(torch.equal(x, y) for x, y in zip(out["single_frame_cvs"], a["single_frame_cvs"]))


def test_two_keyframes_in_flight_equal_sequential_forwards(hip_lib):
    """MonoRecModel.submit keeps 2 keyframes on the GPU at once (separate streams + resident buffers); every
    result must equal the strictly sequential forward of the same keyframe, in eager and in hipGraph mode."""
    batches = [synth.make_batch(1, 64, 96, 2, seed=30 + i) for i in range(6)]
    seq, _ = _model(8, graph=False, in_flight=1)
    with torch.no_grad():
        want = [seq(_to_dev(b))["result"].clone() for b in batches]
    for graph in (False, True):
        pipe, _ = _model(8, graph=graph, in_flight=2)
        got = []
        for rep in range(3 if graph else 1):         # graph mode: warm-up pass, capture pass, replay pass
            got = []
            pending = []
            with torch.no_grad():
                for b in batches:
                    pending.append(pipe.submit(_to_dev(b)))
                    if len(pending) == 2:
                        got.append(pending.pop(0).result()["result"].clone())
                while pending:
                    got.append(pending.pop(0).result()["result"].clone())
            torch.cuda.synchronize()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert torch.equal(g, w)


@pytest.mark.gpu
def test_forward_outputs_are_owned_and_submit_outputs_are_views(hip_lib):
    """monorec_model.py:713-727: every forward allocates its outputs.  `forward()` therefore returns tensors that later
    forwards never touch; `submit()` is the documented zero-copy interface (views of the slot's resident buffers)."""
    from monorec_amd import MonoRecModel
    model = MonoRecModel(cv_depth_steps=8, hip_in_flight=2)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).eval()
    batches = [synth.clone_batch(synth.make_batch(1, 64, 96, 2, seed=70 + i), DEV) for i in range(5)]
    with torch.no_grad():
        outs = [model(dict(b)) for b in batches]
        torch.cuda.synchronize()
        snap = [{k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items()
                 if k in MonoRecModel._OUTPUT_KEYS + ("result", "mask")} for o in outs]
        for b in batches[::-1]:                                         # four more forwards over every slot
            model(dict(b))
        torch.cuda.synchronize()
    for o, s in zip(outs, snap):
        assert o["result"] is o["predicted_inverse_depths"][0] and o["mask"] is o["cv_mask"]     # the reference's aliasing (:723-727)
        for k, v in s.items():
            for a, b in zip(v if isinstance(v, list) else [v], o[k] if isinstance(o[k], list) else [o[k]]):
                assert torch.equal(a, b), k
    assert not torch.equal(outs[0]["result"], outs[2]["result"])        # slot 0 served both keyframes
    with torch.no_grad():                                                # zero-copy interface: same storage two submits later
        h0 = model.submit(dict(batches[0])).synchronize()
        model.submit(dict(batches[1])).synchronize()
        h2 = model.submit(dict(batches[2])).synchronize()
    assert h0["result"].data_ptr() == h2["result"].data_ptr()


@pytest.mark.gpu
def test_data_parallel_wrap_matches_plain_forward(hip_lib):
    """base/base_trainer.py:26-29 / evaluater.py:27-30 wrap the model in nn.DataParallel when n_gpu > 1 (every reference config
    says 8).  Two replicas (both on cuda:0 here - the box has one GPU; the replicas still run from two threads over scattered
    half batches and are gathered) must reproduce the plain batch-2 forward."""
    from monorec_amd import MonoRecModel
    model = MonoRecModel(cv_depth_steps=8)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).eval()
    batch = synth.clone_batch(synth.make_batch(2, 64, 96, 2, seed=3), DEV)
    with torch.no_grad():
        plain = model(dict(batch))
        plain = {k: plain[k].clone() for k in ("result", "cv_mask")}
        wrapped = torch.nn.DataParallel(model, device_ids=[0, 0])
        out = wrapped(dict(batch))
    torch.cuda.synchronize()
    assert out["result"].shape == plain["result"].shape
    for k in ("result", "cv_mask"):
        assert float((out[k] - plain[k]).abs().max()) <= 1e-6, k


def test_inputs_are_read_in_place_or_through_the_resident_copy(hip_lib):
    """Dense fp32 inputs are read where they are (no device copy); non-contiguous / other-dtype inputs go through the slot's
    resident buffers.  Same result either way, and the caller's tensors are never written."""
    model, sd = _model(8, graph=False)
    batch = _to_dev(synth.make_batch(1, 64, 96, 2, seed=12))
    with torch.no_grad():
        want = model(dict(batch))["result"].clone()
        odd = dict(batch)
        odd["keyframe"] = batch["keyframe"].permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)      # same values, strided
        odd["frames"] = [batch["frames"][0].double(), batch["frames"][1]]
        assert not odd["keyframe"].is_contiguous()
        keep = odd["keyframe"].clone()
        got = model(odd)["result"].clone()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(odd["keyframe"], keep)


def test_dynamic_batching_of_a_keyframe_stream(hip_lib):
    """hip_batch_keyframes=K: submit() coalesces K consecutive equal-shaped requests into one launch over their concatenated batch;
    a result asked for before the group is full launches what has been collected.  Every request gets its own slice of the outputs,
    equal (to summation-order noise: other tile schedules at another batch size) to the request run alone."""
    import collections
    plain, sd = _model(8, graph=False)
    batches = [_to_dev(synth.make_batch(1, 64, 96, 2, seed=80 + i)) for i in range(5)]
    with torch.no_grad():
        want = [plain(dict(b))["result"].clone() for b in batches]
    m = MonoRecModel(cv_depth_steps=8, hip_in_flight=2, hip_batch_keyframes=2)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    pending, got = collections.deque(), []
    with torch.no_grad():
        for b in batches:                                   # 5 requests: groups (0,1), (2,3) and the early-flushed (4)
            pending.append(m.submit(dict(b)))
            if len(pending) >= 4:
                got.append(pending.popleft().result()["result"].clone())
        while pending:
            got.append(pending.popleft().result()["result"].clone())
        torch.cuda.synchronize()
        assert len(m._plans) >= 2 and {k[1] for k in m._plans} == {1, 2}      # batch-2 launches plus the single left-over request
        for g, w in zip(got, want):
            assert g.shape == w.shape == (1, 1, 64, 96)
            assert float((g - w).abs().max()) <= 1e-5
        out = m(dict(batches[0]))                           # forward() through a batching model: a group of one, owned outputs
        assert float((out["result"] - want[0]).abs().max()) <= 1e-5 and out["mask"] is out["cv_mask"]


@pytest.mark.gpu
def test_data_parallel_wrap_of_a_fresh_model(hip_lib):
    """evaluater.py:27-30 order: construct, load, .to(device), wrap in nn.DataParallel - and only then the first forward.  The
    replicas torch makes per forward have no `_parameters`; the weight snapshot is taken on the original in
    `_replicate_for_data_parallel` (the wrap used to work only after a plain forward had filled the snapshot)."""
    sd = None
    outs = []
    for wrap in (True, False):
        model = MonoRecModel(cv_depth_steps=8)
        sd = sd or synth.seeded_state_dict(model.state_dict(), seed=0)
        model.load_state_dict(sd)
        model = model.to(DEV).eval()
        batch = synth.clone_batch(synth.make_batch(2, 64, 96, 2, seed=3), DEV)
        assert model._packed_state is None and not model._plans
        runner = torch.nn.DataParallel(model, device_ids=[0, 0]) if wrap else model
        with torch.no_grad():
            out = runner(dict(batch))
        torch.cuda.synchronize()
        outs.append({k: out[k].clone() for k in ("result", "cv_mask")})
    for k in ("result", "cv_mask"):
        assert float((outs[0][k] - outs[1][k]).abs().max()) <= 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 3])
def test_submit_and_result_on_separate_streams(hip_lib, depth):
    """The pipelined loop of bench.py / Evaluater: requests submitted under one stream, results taken - and consumed - under
    another, the host running `hip_queue_depth` forwards per slot ahead.  The submit that reuses a slot must wait for what the
    result stream was given to do with the slot's outputs (here: a reduction enqueued right after result()); every keyframe must
    come out exactly as in a plain sequential loop."""
    import collections
    plain, sd = _model(8, graph=False)
    batches = [_to_dev(synth.make_batch(1, 64, 96, 2, seed=300 + i)) for i in range(9)]
    with torch.no_grad():
        want = [plain(dict(b))["result"].clone() for b in batches]
    torch.cuda.synchronize()
    m = MonoRecModel(cv_depth_steps=8, hip_in_flight=2, hip_queue_depth=depth)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
    pending, got = collections.deque(), []
    big = torch.empty(64 << 20, device=DEV)

    def collect():
        with torch.cuda.stream(s_out):
            out = pending.popleft().result()
            big.normal_()                                   # keeps the result stream busy: the copy below runs LATE
            got.append(out["result"].clone())
    with torch.no_grad():
        for b in batches:
            with torch.cuda.stream(s_in):
                pending.append(m.submit(dict(b)))
            if len(pending) >= 2:
                collect()
        while pending:
            collect()
    torch.cuda.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g, w), i


@pytest.mark.gpu
def test_forward_outputs_are_owned_and_equal_what_submit_shows(hip_lib):
    """forward(): every tensor output (the three constants of :675-677 included) lives in memory of its own - nothing aliases the
    plan's resident buffers - and equals what submit() shows."""
    m, sd = _model(8, graph=False, in_flight=2)
    batch = _to_dev(synth.make_batch(2, 64, 96, 2, seed=21))
    with torch.no_grad():
        view = m.submit(dict(batch)).result()
        keep = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in view.items()
                if k in MonoRecModel._OUTPUT_KEYS}
        out = m(dict(batch))
        resident = {t.data_ptr() for p in m._plans.values() for t in p.buf.values()}
        for _ in range(3):                                   # later forwards reuse both slots
            m(dict(_to_dev(synth.make_batch(2, 64, 96, 2, seed=22))))
    torch.cuda.synchronize()
    assert out["result"] is out["predicted_inverse_depths"][0] and out["mask"] is out["cv_mask"]
    assert float(out["inv_depth_min"]) == float(torch.tensor(0.33)) and int(out["cv_depth_steps"]) == 8 and out["cv_depth_steps"].dtype == torch.int32
    for k, v in keep.items():
        for a, b in zip(out[k] if isinstance(v, list) else [out[k]], v if isinstance(v, list) else [v]):
            assert a.data_ptr() not in resident, k
            assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), k


@pytest.mark.gpu
@pytest.mark.parametrize("host_mats", [False, True])
def test_prepare_then_submit_pipeline(hip_lib, host_mats):
    """The pipelined loop of bench.py / Evaluater: prepare(request i) - pose algebra, with device-side 4x4s one gather launch - while
    the device is busy, host-side wait for the result whose slot request i reuses, submit(request i, token).  Every keyframe equals
    the plain forward; a token is bound to its dict."""
    import collections
    plain, sd = _model(8, graph=False)
    cpu = [synth.make_batch(1, 64, 96, 2, seed=400 + i) for i in range(7)]
    batches = [_to_dev(b) for b in cpu]
    with torch.no_grad():
        want = [plain(dict(b))["result"].clone() for b in batches]
    if host_mats:
        for d, c in zip(batches, cpu):
            for k in ("keyframe_intrinsics", "keyframe_pose", "intrinsics", "poses"):
                d[k] = c[k]
    m = MonoRecModel(cv_depth_steps=8, hip_in_flight=2)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    pending, got = collections.deque(), []
    with torch.no_grad():
        for b in batches:
            req = dict(b)
            token = m.prepare(req)
            if len(pending) >= 2:
                got.append(pending.popleft().synchronize()["result"].clone())
            pending.append(m.submit(req, token))
        while pending:
            got.append(pending.popleft().synchronize()["result"].clone())
        with pytest.raises(ValueError):
            m.submit(dict(batches[0]), m.prepare(dict(batches[0])))
    torch.cuda.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g, w), i


@pytest.mark.gpu
def test_forward_produces_its_outputs_in_two_caller_owned_arenas(hip_lib):
    """forward() of the full model re-targets the launches at memory allocated for the caller (no copy): the maps a caller keeps
    (`result`, the depth scales, `cv_mask`, the three constants) share one small allocation, the volumes and image features another
    - holding on to `result` does not pin the 60 MB of volumes - and a submit() afterwards runs on the resident buffers
    again."""
    m, sd = _model(8, graph=False, in_flight=2)
    batch = _to_dev(synth.make_batch(1, 64, 96, 2, seed=31))
    with torch.no_grad():
        out = m(dict(batch))
        plan = next(iter(m._plans.values()))
        # the launches were re-targeted at the caller's arenas while they were enqueued and the plan is back on its resident buffers as soon as
        # forward() returns (anything that drives the plan directly afterwards must not write into memory the caller may free)
        assert plan.outputs_rebindable and plan.bound == plan._resident
        assert out["result"].data_ptr() not in {t.data_ptr() for t in plan.buf.values()} and out["cost_volume"].data_ptr() != plan.buf["cost_volume"].data_ptr()
        small = out["result"].untyped_storage()
        assert small.data_ptr() == out["cv_mask"].untyped_storage().data_ptr() == out["inv_depth_min"].untyped_storage().data_ptr()
        assert small.data_ptr() == out["predicted_inverse_depths"][3].untyped_storage().data_ptr()
        big = out["cost_volume"].untyped_storage()
        assert big.data_ptr() != small.data_ptr() and big.data_ptr() == out["image_features"][4].untyped_storage().data_ptr()
        assert big.data_ptr() == out["single_frame_cvs"][1].untyped_storage().data_ptr()
        assert small.nbytes() < 64 * 96 * 4 * 4 and big.nbytes() > small.nbytes()
        res = out["result"].clone()
        view = m.submit(dict(batch)).synchronize()          # slot 0 again (the counter did not move): resident buffers
        assert plan.bound == plan._resident
        assert view["result"].data_ptr() == plan.buf["pred0"].data_ptr() and torch.equal(view["result"], res)
        out2 = m(dict(batch))
        torch.cuda.synchronize()
        assert out2["result"].data_ptr() != view["result"].data_ptr() and torch.equal(out2["result"], res)
        for k in ("cost_volume", "cv_mask"):
            assert torch.equal(out2[k], view[k]), k
        for a, b in zip(out2["image_features"] + out2["single_frame_cvs"], view["image_features"] + view["single_frame_cvs"]):
            assert torch.equal(a, b)
    assert float(out["inv_depth_max"]) == float(torch.tensor(0.0025)) and int(out["cv_depth_steps"]) == 8


@pytest.mark.gpu
def test_forward_on_the_callers_stream_equals_the_slot_stream_path(hip_lib):
    """Round 6: forward() enqueues the keyframe on the CALLER'S stream (default) - no host wait, the next call's gather and encoder
    stage queue up behind the running forward.  A run of forwards with nothing between them (the host runs ahead), on the default stream and on a
    side stream, with matrices on the device and on the host, must give bit for bit what the slot-stream path of rounds 3-5
    (hip_forward_on_callers_stream=False) gives; the outputs stay owned; a submit() on the same slot afterwards is ordered behind them."""
    batches = [synth.make_batch(1, 64, 96, 2, seed=40 + i) for i in range(4)]
    results = {}
    for inline in (True, False):
        m, sd = _model(8, graph=False, hip_forward_on_callers_stream=inline)
        outs = []
        with torch.no_grad():
            devb = [_to_dev(b) for b in batches]
            for k in ("keyframe_intrinsics", "keyframe_pose", "intrinsics", "poses"):      # one request with its 4x4s on the host
                devb[1][k] = batches[1][k]
            torch.cuda.synchronize()
            for b in devb[:3]:
                outs.append(m(dict(b)))                                                    # no synchronisation in between
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                outs.append(m(dict(devb[3])))
                outs.append(m(dict(devb[0])))
            torch.cuda.current_stream().wait_stream(side)
            plan = next(iter(m._plans.values()))
            resident = {t.data_ptr() for t in plan.buf.values()}
            assert all(o["result"].data_ptr() not in resident for o in outs)
            view = m.submit(dict(devb[2])).synchronize()                                   # slot 0 on its own stream: behind the forwards
            torch.cuda.synchronize()
            assert torch.equal(view["result"], outs[2]["result"])
        results[inline] = [{k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in o.items()
                            if k in ("result", "cv_mask", "cost_volume", "single_frame_cvs", "image_features", "predicted_inverse_depths")} for o in outs]
    for a, b in zip(results[True], results[False]):
        for k in a:
            if torch.is_tensor(a[k]):
                assert torch.equal(a[k], b[k]), k
            else:
                assert all(torch.equal(x, y) for x, y in zip(a[k], b[k])), k
    assert torch.equal(results[True][0]["result"], results[True][4]["result"])             # the same request on the two streams


@pytest.mark.gpu
@pytest.mark.parametrize("pretrain_mode", [0, 1])
def test_forward_between_a_submit_and_its_result_leaves_the_handle_intact(hip_lib, pretrain_mode):
    """submit(a); model(b); a.result() - the forward must not overwrite the outputs the pending handle views.  The full
    model (pretrain_mode 0) writes forward()'s outputs into caller-owned memory; a plan that copies out of its resident buffers
    (pretrain_mode 1: constant zero mask) moves the forward to a slot without an uncollected handle, and raises when there is none."""
    def build(in_flight):
        m = MonoRecModel(cv_depth_steps=8, hip_in_flight=in_flight, pretrain_mode=pretrain_mode)
        m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=0))
        return m.to(DEV).eval()
    a = _to_dev(synth.make_batch(1, 64, 96, 2, seed=41))
    b = _to_dev(synth.make_batch(1, 64, 96, 2, seed=42))
    m = build(2)
    with torch.no_grad():
        want_a = m(dict(a))["result"].clone()
        want_b = m(dict(b))["result"].clone()
        assert not torch.equal(want_a, want_b)
        m._slot_counter[0] = 0
        ha = m.submit(dict(a))                               # slot 0, not collected
        got_b = m(dict(b))["result"]
        got_a = ha.result()["result"]
        torch.cuda.synchronize()
        assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
    if pretrain_mode == 1:
        m1 = build(1)
        with torch.no_grad():
            h = m1.submit(dict(a))
            with pytest.raises(RuntimeError, match="result has not been taken"):
                m1(dict(b))
            h.result()
            assert torch.equal(m1(dict(b))["result"], want_b)


@pytest.mark.gpu
def test_wrongly_shaped_pose_matrices_raise(hip_lib):
    """the gather launch of prepare() reads 16 * batch floats per matrix - an unbatched (4, 4) matrix must raise, not overrun."""
    m, _ = _model(8, graph=False)
    batch = _to_dev(synth.make_batch(2, 64, 96, 2, seed=43))
    bad = dict(batch)
    bad["keyframe_pose"] = batch["keyframe_pose"][0]
    with torch.no_grad(), pytest.raises(ValueError, match="4, 4"):
        m(bad)
    bad = dict(batch)
    bad["poses"] = [batch["poses"][0], batch["poses"][1][:1]]
    with torch.no_grad(), pytest.raises(ValueError, match="4, 4"):
        m.prepare(bad)


# ---- the second, ill-conditioned weight family: every parity bar again, unchanged ---------------------
@pytest.mark.parametrize("exact", [False, "f2", True])
def test_harsh_weights_small_case_under_every_conv_form_setting(hip_lib, exact):
    """synth family "harsh" (9x channel-scale range, near-cancelling filters, BatchNorm variances down to 1e-3) on the small case:
    the committed table (F(4,.) / F(4x4,3x3) forms where it selects them), `hip_exact_convs="f2"` (F(2,.) forms only) and
    `hip_exact_convs=True` (direct kernel only) all inside the same bars against the reference fixture and the oracle on this host."""
    g = Golden("small_harsh")
    assert g.family == "harsh"
    model, sd = _model(g.depths, graph=False, family="harsh", hip_exact_convs=exact)
    batch = g.make_inputs()
    with torch.no_grad():
        out = model(_to_dev(batch))
    torch.cuda.synchronize()
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=g.depths), f"small_harsh exact={exact}")
    info = g.compare("result", out["result"], atol=RESULT_ATOL)
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    for i in range(4):
        g.compare(f"pred{i}", out["predicted_inverse_depths"][i], atol=RESULT_ATOL)
    for i in range(5):
        g.compare(f"feat{i}", out["image_features"][i], atol=2e-4, rtol=1e-4)
    print(f"small_harsh exact={exact}: result vs reference fixture", info)
    plan = next(iter(model._plans.values()))
    forms = {(c.get("wino_m", 2) if min(c["k"]) == 1 else (4 if c.get("wino_variant") == 3 else 2)) for c in plan.conv_log if c.get("winograd")}
    assert forms <= ({2, 4} if exact is False else ({2} if exact == "f2" else set())), forms


def test_harsh_weights_c2_config_against_the_reference_fixture(hip_lib):
    """BASELINE configs[1] with the ill-conditioned family and the COMMITTED table (F(4,7) on depth.enc0.0, F(4,3) on six layers,
    F(4x4,3x3) on mask.enc0.*): depth within 1e-4 of the reference's CPU output; the exact-path switch for comparison."""
    g = Golden("c1_256x512_harsh")
    batch = g.make_inputs()
    errs = {}
    for exact in (False, True):
        model, sd = _model(g.depths, graph=False, family="harsh", hip_exact_convs=exact)
        with torch.no_grad():
            out = model(_to_dev(batch))
        torch.cuda.synchronize()
        errs[exact] = g.compare("result", out["result"], atol=RESULT_ATOL)["max_abs"]
        g.compare("cv_mask", out["cv_mask"], atol=1e-4)
        g.compare("cost_volume", out["cost_volume"], atol=1e-4, max_outlier_frac=5e-4)
        for i in range(5):
            g.compare(f"feat{i}", out["image_features"][i], atol=2e-4, rtol=1e-4)
        for i in range(4):
            g.compare(f"pred{i}", out["predicted_inverse_depths"][i], atol=RESULT_ATOL)
        if exact is False:
            plan = next(iter(model._plans.values()))
            assert any(c.get("wino_m") == 4 for c in plan.conv_log), "the committed table no longer selects an F(4,.) form at c2"
        del model
    print("c2 harsh weights: result vs reference fixture, table %.2e / direct kernel only %.2e" % (errs[False], errs[True]))


def test_harsh_weights_c3_full_shape_against_the_oracle(hip_lib):
    model, sd = _model(64, graph=False, family="harsh")
    batch = synth.make_batch(8, 256, 512, 4, seed=3)
    with torch.no_grad():
        out = model(_to_dev(batch))
        out = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in out.items() if k in
               ("result", "cv_mask", "predicted_inverse_depths", "image_features", "cost_volume", "single_frame_cvs")}
    torch.cuda.synchronize()
    _check_against(out, orc.forward(sd, batch, cv_depth_steps=64), "c3 full shape, harsh weights")


def test_harsh_weights_on_the_reference_example_sample_stage_by_stage(hip_lib):
    """The real KITTI sample with the ill-conditioned family: every stage downstream of this run's own cost volume / features at 1e-4
    against the oracle's modules (the end-to-end statement on real data is test_reference_example_sample_with_the_fixtures_own_matrices)."""
    g = Golden("kitti_example_169")
    batch = g.make_inputs()
    model, sd = _model(g.depths, graph=False, family="harsh")
    with torch.no_grad():
        out = model(_to_dev(batch))
    torch.cuda.synchronize()
    feats = [t.cpu() for t in out["image_features"]]
    want_feats = orc.resnet_features(sd, batch["keyframe"] + 0.5)
    for i in range(5):
        rel = ((feats[i] - want_feats[i]).abs().max() / want_feats[i].abs().max().clamp_min(1e-6)).item()
        assert rel <= 1e-4, (i, rel)
    mask_ref = orc.mask_module(sd, [t.cpu() for t in out["single_frame_cvs"]], feats)
    assert (out["cv_mask"].cpu() - mask_ref).abs().max().item() <= 1e-4
    preds = orc.depth_module(sd, out["cost_volume"].cpu(), batch["keyframe"], feats)
    for i in range(4):
        want = (1 - preds[i]) * 0.0025 + preds[i] * 0.33
        err = (out["predicted_inverse_depths"][i].cpu() - want).abs().max().item()
        print("kitti example, harsh weights: depth scale %d vs the oracle's depth module on this run's volume: %.2e" % (i, err))
        assert err <= RESULT_ATOL, (i, err)


@pytest.mark.gpu
def test_the_models_streams_are_created_once_in_a_fixed_order(hip_lib):
    """Where the model's four busy streams sit among ROCm's hardware queues - bound at first use, in order of first use - sets the two-keyframes-in-flight rate
    for the life of the process (tools/sessions/r05_s4.sh - s7.sh: next to each other 757-769 keyframes/s at c2, something between them 693-717, spread out
    509-558), so MonoRecModel._device_streams creates AND first uses ALL of them at one point in one order - the slots' main streams, their encoder streams,
    the gather stream - whatever the caller does first: prepare() + submit(), a bare submit(), or forward() (which launches on a slot's encoder stream
    before its main stream: the first-use order that cost the first tree of round 5 8 %)."""
    batch = _to_dev(synth.make_batch(1, 64, 96, 2, seed=77))
    orders = []
    for first in ("prepare", "submit", "forward"):
        m, _ = _model(8, graph=False, in_flight=2, hip_slot_streams=2)
        with torch.no_grad():
            if first == "prepare":
                req = dict(batch)
                tok = m.prepare(req)
                m.submit(req, tok).synchronize()
            elif first == "submit":
                m.submit(dict(batch)).synchronize()
            else:
                m(dict(batch))
        torch.cuda.synchronize()
        ds = m._dev_streams[str(torch.device(DEV))]
        assert [k for k in ds if k != "_pads"] == ["m0", "m1", "e0", "e1", "g"]
        assert m._slot_streams(0, torch.device(DEV))["main"] is ds["m0"] and m._slot_streams(1, torch.device(DEV))["enc"] is ds["e1"]
        assert m._slot_streams(1, torch.device(DEV), own=True)["enc"] is ds["e1"]
        assert all(v[1] is ds["g"] for v in m._prep_pinned.values())
        ids = [ds[k].cuda_stream for k in ("g", "m0", "e0", "m1", "e1")]
        assert len(set(ids)) == 5                                # five distinct streams, each first used at creation, in this order
        orders.append(first)
    assert orders == ["prepare", "submit", "forward"]
    # the default since round 5: four slots with ONE stream each (820-827 keyframes/s at c2 against 762-766 for two slots x two streams, r05_s9); only
    # forward() - one keyframe at a time - keeps its encoder stage on a second stream: the next slot's
    m, _ = _model(8, graph=False, in_flight=4)
    with torch.no_grad():
        m.submit(dict(batch)).synchronize()
    ds = m._dev_streams[str(torch.device(DEV))]
    assert [k for k in ds if k != "_pads"] == ["m0", "m1", "m2", "m3", "g"]
    sub, own = m._slot_streams(2, torch.device(DEV)), m._slot_streams(2, torch.device(DEV), own=True)
    assert sub["enc"] is sub["main"] is ds["m2"] and own["main"] is ds["m2"] and own["enc"] is ds["m3"]
    assert m._slot_streams(3, torch.device(DEV), own=True)["enc"] is ds["m0"]
    # one slot: two streams (the encoder stage beside the cost volume), as in rounds 2-4
    m, _ = _model(8, graph=False, in_flight=1)
    with torch.no_grad():
        m(dict(batch))
    ds = m._dev_streams[str(torch.device(DEV))]
    assert [k for k in ds if k != "_pads"] == ["m0", "e0", "g"] and m._slot_streams(0, torch.device(DEV))["enc"] is ds["e0"]
