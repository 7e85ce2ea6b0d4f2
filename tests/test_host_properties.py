"""Property tests (hypothesis) of the host-side logic that runs without a GPU: the C coefficient-table function of the
resize, crop box / intrinsics, the Evaluater bookkeeping and the Pillow restatement itself against the installed Pillow."""
import ctypes
import math

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from monorec_amd import _lib, evaluate, input_pipeline
from oracle import input_oracle
from oracle.kitti_oracle import intrinsics_matrix, target_intrinsics

FAST = settings(max_examples=120, deadline=None)


@FAST
@given(n_in=st.integers(1, 2000), n_out=st.integers(1, 1200), lo=st.integers(0, 40), cut=st.integers(0, 40))
def test_coefficient_tables_for_any_size_and_box(hip_lib, n_in, n_out, lo, cut):
    """mr_resample_coeffs_bilinear (double arithmetic of Pillow's precompute_coeffs + normalize_coeffs_8bpc) == restatement."""
    in0 = min(lo, n_in - 1)
    in1 = max(in0 + 1, n_in - cut)
    ks = int(hip_lib.mr_resample_ksize_bilinear(in0, in1, n_out))
    ks_o, b_o, c_o = input_oracle.resample_coeffs(n_in, in0, in1, n_out)
    assert ks == ks_o
    bounds = np.zeros((n_out, 2), dtype=np.int32)
    coeffs = np.zeros((n_out, ks), dtype=np.int32)
    _lib.check(hip_lib.mr_resample_coeffs_bilinear(n_in, in0, in1, n_out, bounds.ctypes.data, coeffs.ctypes.data), "coeffs")
    assert np.array_equal(bounds, b_o) and np.array_equal(coeffs, c_o)
    assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all()


@FAST
@given(h=st.integers(8, 1300), w=st.integers(8, 1300), th=st.integers(1, 20), tw=st.integers(1, 20),
       fx=st.floats(100, 1500), cx=st.floats(0, 1), cy=st.floats(0, 1))
def test_crop_box_and_intrinsics_match_the_restatement(h, w, th, tw, fx, cx, cy):
    target = (32 * th, 32 * tw)
    p = np.array([[fx, 0, cx * w, 3.0], [0, fx * 1.01, cy * h, 0.1], [0, 0, 1, 0.005]])
    fr, box = input_pipeline.compute_target_intrinsics(p, (h, w), target)
    assert tuple(box) == tuple(input_oracle.crop_box_for(h, w, *target))
    assert fr == target_intrinsics(p, (h, w), target)                             # same double expressions, bit for bit
    assert torch.equal(input_pipeline.format_intrinsics(fr, target), intrinsics_matrix(fr, target))
    x0, y0, x1, y1 = box
    assert x0 >= 0 and y0 >= 0 and x1 <= w and y1 <= h and float(x0).is_integer() and float(y1).is_integer()


@settings(max_examples=40, deadline=None)
@given(h=st.integers(3, 60), w=st.integers(3, 60), oh=st.integers(1, 50), ow=st.integers(1, 50), c=st.sampled_from([1, 3]),
       seed=st.integers(0, 10_000))
def test_resize_restatement_equals_the_installed_pillow(h, w, oh, ow, c, seed):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, size=(h, w, c) if c == 3 else (h, w)).astype(np.uint8)
    want = np.array(Image.fromarray(img).resize((ow, oh), resample=Image.BILINEAR))
    assert np.array_equal(input_oracle.resize_bilinear_u8(img, oh, ow), want)


@FAST
@given(data=st.data(), n_metrics=st.integers(1, 7), n_batches=st.integers(1, 12))
def test_evaluation_log_equals_the_written_out_loop(data, n_metrics, n_batches):
    """evaluater.py:94-118: a NaN anywhere invalidates the batch; plain mean over valid batches; size-weighted running mean."""
    value = st.one_of(st.floats(0, 100), st.just(float("nan")))
    per_batch = [[data.draw(value) for _ in range(n_metrics)] for _ in range(n_batches)]
    sizes = [data.draw(st.integers(1, 4)) for _ in range(n_batches)]
    log = evaluate.evaluation_log(per_batch, sizes)
    tot, valid, run, num = np.zeros(n_metrics), 0, np.zeros(n_metrics), 0
    for m, bs in zip(per_batch, sizes):
        acc = np.array(m, dtype=np.float64)
        ok = not np.isnan(acc).any()
        acc = acc if ok else np.zeros(n_metrics)
        tot, valid = tot + acc, valid + (1 if ok else 0)
        run = acc.copy() if num == 0 else run * (num / (num + bs)) + acc * (bs / (num + bs))
        num += bs
    assert log["valid_batches"] == valid
    if valid:
        assert np.allclose(log["metrics"], tot / valid, rtol=1e-12, atol=0)
    else:
        assert all(math.isnan(v) or math.isinf(v) for v in log["metrics"])
    assert np.allclose(log["metrics_correct"], run, rtol=1e-12, atol=0)
