"""GPU (MI355X): the fused sparse-metric reduction (mr_sparse_metric_sums_f32 behind monorec_amd.metrics) against the
CPU oracle and the committed outputs of the reference's own metric functions (tests/golden/sparse_metrics.json)."""
import json
import os

import pytest
import torch

from golden_util import GOLDEN
from monorec_amd import metrics, synth
from oracle import monorec_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 2e-5     # fp32 per-element terms, fp64 accumulation here vs fp32 torch.sum in the reference


@pytest.mark.parametrize("name", ["default_eval", "roi_no_maxdist", "full_size"])
def test_sparse_metrics_match_reference_fixture_and_oracle(hip_lib, name):
    case = json.load(open(os.path.join(GOLDEN, "sparse_metrics.json")))[name]
    b, h, w, seed, roi, maxd = case["config"]
    pred, gt = synth.make_depth_pair(b, h, w, seed)
    data = {"result": pred.to(DEV), "target": gt.to(DEV)}
    ref = orc.sparse_metrics(pred, gt, roi, maxd)
    for fn in metrics.SPARSE_METRICS:
        got = float(getattr(metrics, fn)(data, roi, maxd))
        want = case["metrics"][fn]
        assert abs(got - want) <= RTOL * max(1.0, abs(want)), (fn, got, want)
        assert abs(got - float(ref[fn])) <= RTOL * max(1.0, abs(want)), (fn, got, float(ref[fn]))
    assert metrics._CACHE_KEY in data            # the seven calls shared one launch


def test_metric_edge_cases(hip_lib):
    pred, gt = synth.make_depth_pair(2, 32, 64, 3)
    gt[1] = 0                                      # a sample without any valid ground truth -> NaN like the reference
    data = {"result": pred.to(DEV), "target": gt.to(DEV)}
    ref = orc.sparse_metrics(pred, gt, None, 80)
    assert torch.isnan(metrics.rmse_sparse_metric(data, None, 80)) and torch.isnan(ref["rmse_sparse_metric"])
    got, want = float(metrics.abs_rel_sparse_metric(data, None, 80)), float(ref["abs_rel_sparse_metric"])
    assert abs(got - want) <= RTOL * max(1.0, abs(want))
    # predictions below 1/max_distance and negative ones are clamped exactly like the reference
    pred2 = pred.clone()
    pred2[0, 0, :4] = -0.1
    pred2[0, 0, 4:8] = 1e-4
    data2 = {"result": pred2.to(DEV), "target": gt.to(DEV)}
    ref2 = orc.sparse_metrics(pred2, gt, None, 80)
    got2 = float(metrics.sq_rel_sparse_metric(data2, None, 80))
    assert abs(got2 - float(ref2["sq_rel_sparse_metric"])) <= RTOL * max(1.0, abs(float(ref2["sq_rel_sparse_metric"])))


def test_metrics_on_model_output(hip_lib):
    """result of the HIP model -> metrics on device, against the all-CPU oracle chain."""
    from monorec_amd import MonoRecModel
    model = MonoRecModel(cv_depth_steps=8, hip_in_flight=1)
    sd = synth.seeded_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    batch = synth.make_batch(2, 64, 96, 2, seed=1)
    _, gt = synth.make_depth_pair(2, 64, 96, 11)
    with torch.no_grad():
        data = model(synth.clone_batch(batch, DEV))
    data["target"] = gt.to(DEV)
    ref_out = orc.forward(sd, batch, cv_depth_steps=8)
    ref = orc.sparse_metrics(ref_out["result"], gt, None, 80)
    for fn in metrics.SPARSE_METRICS:
        got, want = float(getattr(metrics, fn)(data, None, 80)), float(ref[fn])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (fn, got, want)


@pytest.mark.parametrize("name", ["flags_eval", "flags_roi"])
def test_onlyvalid_and_onlydynamic_variants_match_the_reference_fixture(hip_lib, name):
    """pred_all_valid=False (`*_sparse_onlyvalid_metric`) and use_cvmask=True (`*_sparse_onlydynamic_metric`) - sparse_metrics.py:81-212,
    utils/util.py:101-107 - against the committed outputs of the reference's own functions (oracle/make_golden_metric_flags.py)."""
    BASES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
    case = json.load(open(os.path.join(GOLDEN, "sparse_metrics_flags.json")))[name]
    b, h, w, seed, roi, maxd = case["config"]
    pred, gt, mv = synth.make_metric_flag_inputs(b, h, w, seed)
    assert float((pred == 0).float().mean()) > 0.05
    data = {"result": pred.to(DEV), "target": gt.to(DEV), "mvobj_mask": mv.to(DEV)}
    checked = 0
    for key, want in case["metrics"].items():
        if key.endswith("_both"):
            got = float(getattr(metrics, key[:-5] + "_sparse_metric")(data, roi, maxd, False, True))
        else:
            got = float(getattr(metrics, key)(data, roi, maxd))
        assert abs(got - want) <= RTOL * max(1.0, abs(want)), (key, got, want)
        checked += 1
    assert checked == (21 if roi is None else 7) and set(k.split("_sparse_")[0] for k in case["metrics"] if "_sparse_" in k) == set(BASES)
    if roi is not None:
        with pytest.raises(RuntimeError):
            metrics.a1_sparse_onlydynamic_metric(data, roi, maxd)
    plain = float(metrics.a1_sparse_metric(data, roi, maxd))
    assert abs(plain - case["metrics"]["a1_sparse_onlyvalid_metric"]) > 1e-4        # the option does change the value on this input
