"""CPU: the oracle restatement against the committed fixtures (= outputs of the real reference,
written by oracle/make_golden.py in the build container)."""
import json
import os

import pytest
import torch

from golden_util import GOLDEN, Golden
from monorec_amd import synth
from monorec_amd.model import MonoRecModel
from oracle import monorec_oracle as orc

# The fixtures were produced on the build container's CPU; on another host MKL/oneDNN may order sums
# differently, so allow fp32 noise (reference's own 1-vs-8-thread noise: 6e-8) and the
# rare validity-mask flip that a 1-ulp projection difference can cause.
ATOL = 2e-5
FLIPS = 2e-4


def test_pinning_report_says_oracle_equals_reference():
    rep = json.load(open(os.path.join(GOLDEN, "PINNING.json")))
    assert all(rep["low_level"].values())
    for case, info in rep["cases"].items():
        assert max(info["oracle_vs_reference_maxabs"].values()) == 0.0, case


@pytest.mark.parametrize("case", ["small", "small_hard_pose", "d64_f4", "small_harsh"])
def test_oracle_full_model_matches_reference_fixture(case):
    """(`small_harsh`: the ill-conditioned weight family of synth.seeded_state_dict, fixture by oracle/make_golden_harsh.py)"""
    g = Golden(case)
    model = MonoRecModel(cv_depth_steps=g.depths)
    sd = synth.seeded_state_dict(model.state_dict(), seed=0, family=g.family)
    out = orc.forward(sd, g.make_inputs(), cv_depth_steps=g.depths)
    g.compare("result", out["result"], atol=ATOL)
    g.compare("cv_mask", out["cv_mask"], atol=ATOL)
    g.compare("cost_volume", out["cost_volume"], atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(out["single_frame_cvs"]):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(out["image_features"]):
        g.compare(f"feat{i}", t, atol=1e-4, rtol=1e-5)
    for i, t in enumerate(out["predicted_inverse_depths"]):
        g.compare(f"pred{i}", t, atol=ATOL)


def test_oracle_on_the_reference_example_sample():
    """KITTI seq 07 image 169 (the reference's example/test_monorec.py sample, read through the reference's own
    dataset class when the fixture was made): real images, DVSO poses (translation ~80 m), annotated lidar depth."""
    g = Golden("kitti_example_169")
    batch = g.make_inputs()
    assert batch["keyframe"].shape == (1, 3, 256, 512) and len(batch["frames"]) == 2
    model = MonoRecModel(cv_depth_steps=g.depths)
    sd = synth.seeded_state_dict(model.state_dict(), seed=0)
    out = orc.forward(sd, batch, cv_depth_steps=g.depths)
    g.compare("result", out["result"], atol=1e-4)
    g.compare("cv_mask", out["cv_mask"], atol=1e-4)
    g.compare("cost_volume", out["cost_volume"], atol=2e-4, max_outlier_frac=5e-4)
    got = orc.sparse_metrics(torch.from_numpy(g.z["result.full"]), g.target(), None, 80)
    for k, want in g.reference_metrics().items():
        assert abs(float(got[k]) - want) <= 2e-6 * max(1.0, abs(want)), (k, float(got[k]), want)


def test_oracle_cost_volume_ragged_size():
    g = Golden("cv_only_ragged")
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths)
    g.compare("cost_volume", cv, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(sf):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)


def test_cost_volume_range_and_invalid_pixels():
    g = Golden("small")
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths)
    assert cv.min() >= -1 and cv.max() <= 1
    # the 2 px border is always invalid (border mask, monorec_model.py:282-284)
    assert float(cv[:, :, :2].abs().max()) == 0 and float(cv[:, :, :, -2:].abs().max()) == 0
    for t in sf:
        assert float(t[:, :, :, :2].abs().max()) == 0


def test_seeded_weights_are_order_independent():
    m = MonoRecModel(cv_depth_steps=8)
    a = synth.seeded_state_dict(m.state_dict(), seed=0)
    b = synth.seeded_state_dict(dict(reversed(list(m.state_dict().items()))), seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_oracle_sparse_metrics_match_reference_fixture():
    fx = json.load(open(os.path.join(GOLDEN, "sparse_metrics.json")))
    for name, case in fx.items():
        b, h, w, seed, roi, maxd = case["config"]
        pred, gt = synth.make_depth_pair(b, h, w, seed)
        got = orc.sparse_metrics(pred, gt, roi, maxd)
        for k, want in case["metrics"].items():
            assert abs(float(got[k]) - want) <= 2e-6 * max(1.0, abs(want)), (name, k, float(got[k]), want)


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_oracle_use_ssim_variants_match_reference_fixture(mode):
    g = Golden(f"cv_ssim{mode}")
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths, use_ssim=(False if mode == 0 else mode))
    g.compare("cost_volume", cv, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(sf):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)


def test_oracle_per_pixel_depths_match_reference_fixture():
    g = Golden("cv_pixel_depths")
    pix = synth.make_pixel_depths(g.batch, g.depths, g.h, g.w, seed=34)
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths, cv_depths=pix)
    g.compare("cost_volume", cv, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(sf):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)


def test_oracle_without_mult_mask_matches_reference_fixture():
    g = Golden("cv_no_mult_mask")
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths, sfcv_mult_mask=False)
    g.compare("cost_volume", cv, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(sf):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)


@pytest.mark.parametrize("patch", [1, 5, 7])
def test_oracle_patch_sizes_match_reference_fixture(patch):
    g = Golden(f"cv_patch{patch}")
    cv, sf = orc.cost_volume(g.make_inputs(), steps=g.depths, patch_size=patch)
    g.compare("cost_volume", cv, atol=ATOL, max_outlier_frac=FLIPS)
    for i, t in enumerate(sf):
        g.compare(f"sfcv{i}", t, atol=ATOL, max_outlier_frac=FLIPS)


OPTION_CASES = {"pm1": dict(pretrain_mode=1), "pm2": dict(pretrain_mode=2), "pm3": dict(pretrain_mode=3),
                "nocv": dict(no_cv=True), "mask_nocv": dict(mask_use_cv=False), "mask_nofeats": dict(mask_use_feats=False),
                "simple": dict(simple_mask=True)}


@pytest.mark.parametrize("case", sorted(OPTION_CASES))
def test_oracle_model_options_match_reference_fixture(case):
    """pretrain_mode 1/2/3 (eval), no_cv, mask_use_cv / mask_use_feats = False: fixture written from the reference model."""
    from monorec_amd import MonoRecModel
    g = Golden("small_options")
    kw = OPTION_CASES[case]
    batch = g.make_inputs()
    batch["mvobj_mask"] = torch.from_numpy(g.z["input.mvobj_mask"])
    if case == "simple":                      # SimpleMaskModule reads a previous prediction from the dict (monorec_model.py:453)
        batch["predicted_inverse_depths"] = [torch.from_numpy(g.z["input.prev_depth"])]
    sd = synth.seeded_state_dict(MonoRecModel(cv_depth_steps=g.depths, **kw).state_dict(), seed=0)
    out = orc.forward(sd, batch, cv_depth_steps=g.depths, **kw)
    g.compare(f"{case}.result", out["result"], atol=ATOL)
    g.compare(f"{case}.cv_mask", out["cv_mask"], atol=ATOL)
    g.compare(f"{case}.cost_volume", out["cost_volume"], atol=ATOL, max_outlier_frac=FLIPS)
    assert ("predicted_inverse_depths" in out) == (case != "pm2") and ("mask" in out) == (case != "pm2")


def test_conv3d_of_the_box_stage_is_one_fma_chain_position_major():
    """Low-level pin used by DESIGN 2: `F.conv3d(diff, sad_kernel)` (monorec_model.py:247) accumulates its 27 products as ONE
    fused-multiply-add chain, tap position major (row-major over the 3x3 window), channel minor, starting with a plain product -
    bit for bit on the hosts the fixtures come from (oneDNN, AVX-512).  The kernels sum channels first and positions second,
    which is where their 2-5e-7 distance to the reference's single-frame volumes comes from."""
    import torch.nn.functional as F
    from golden_util import Golden
    g = Golden("small")
    batch = g.make_inputs()
    st = {}
    orc.cost_volume(batch, steps=g.depths, stages=st)
    cw = torch.tensor([5 / 32, 16 / 32, 11 / 32]) / 9
    exact = True
    for n in range(g.batch):
        sad, warped = st["sad"][n], st["warped"][n]                    # (F,D,H,W), (D,F,C,H,W)
        d, nf, c, h, w = warped.shape
        diff = orc.ssim_distance(warped.reshape(d * nf, c, h, w) + .5,
                                 batch["keyframe"][n].unsqueeze(0).expand(d * nf, -1, -1, -1) + .5).view(d, nf, c, h, w)
        dp = F.pad(diff, (1, 1, 1, 1)).double()
        acc = None
        for ky in range(3):
            for kx in range(3):
                for ch in range(3):
                    v = dp[:, :, ch, ky:ky + h, kx:kx + w]
                    acc = (v.float() * cw[ch]) if acc is None else (v * cw[ch].double() + acc.double()).float()   # fma: one rounding
        exact = exact and bool((acc.permute(1, 0, 2, 3) == sad).all())
    if not exact:
        pytest.skip("this host's conv3d accumulates in another order (the pin holds on the fixture-generating AVX-512 / oneDNN hosts)")


def test_harsh_weight_family_is_ill_conditioned_but_keeps_the_layer_gain():
    """synth.seeded_state_dict(family="harsh") - the second weight family of the numerics gate: a 9x range of output
    channel scales, near-cancelling alternating-sign filters in every layer with >= 3 taps, BatchNorm variances down to 1e-3 - at the
    He family's RMS per layer, so that the heads stay out of saturation and the 1e-4 depth bar stays meaningful."""
    import math
    tmpl = MonoRecModel(cv_depth_steps=32).state_dict()
    he, harsh = synth.seeded_state_dict(tmpl, 0), synth.seeded_state_dict(tmpl, 0, family="harsh")
    assert set(he) == set(harsh)
    assert all(torch.equal(he[k], synth.seeded_state_dict(tmpl, 0, family="he")[k]) for k in list(he)[:40])
    k = "depth_module.enc.0.0.conv_y.weight"                     # 7 x 1, the layer the table runs as F(4,7)
    w = harsh[k]
    assert abs(float(w.pow(2).mean().sqrt() / he[k].pow(2).mean().sqrt()) - 1.0) < 0.03     # normalised to the uniform law's RMS, bound / sqrt(3)
    rms = w.pow(2).mean(dim=(1, 2, 3)).sqrt()
    assert float(rms.max() / rms.min()) > 5.0                    # channel scales spread over most of [1/3, 3] (x the cancelling filters' 3x)
    f = w[3]                                                     # a near-cancelling filter: alternating taps, sum ~ 2 % of the taps' size
    assert float(f.sum(dim=1).abs().max()) < 0.05 * float(f.abs().sum(dim=1).min()) * 7 and bool((f[:, 0] * f[:, 1] < 0).all())
    var = harsh["_feature_extractor.encoder.layer1.0.bn1.running_var"]
    assert float(var.min()) < 0.05 and float(var.max()) > 0.5
    # folded BatchNorm scale: same RMS as the He family, far wider range
    fold = lambda sd_: sd_["_feature_extractor.encoder.layer1.0.bn1.weight"] / torch.sqrt(sd_["_feature_extractor.encoder.layer1.0.bn1.running_var"] + 1e-5)
    a, b = fold(he), fold(harsh)
    assert abs(float(b.pow(2).mean().sqrt() / a.pow(2).mean().sqrt()) - 1.0) < 1e-3 and float(b.max() / b.min()) > 3 * float(a.max() / a.min())
    with pytest.raises(ValueError):
        synth.seeded_state_dict(tmpl, 0, family="nope")
