"""Sparse depth metrics of the reference (model/metric_functions/sparse_metrics.py:136-252), same function names
and signatures, computed by ONE fused HIP reduction (mr_sparse_metric_sums_f32) and one 64*B byte device->host
copy per batch instead of ~70 full-tensor ATen passes and 7 host syncs (evaluater/evaluater.py:38-50).

`evaluate.py` looks metrics up by name (`getattr(module_metric, met)`, evaluate.py:24): bind this module instead of
`model.metric` to use them.  The seven calls of one batch share a single kernel launch (the sums are cached on the
data dict).  The two options the evaluation configs never set - `pred_all_valid=False` (utils/util.py:105-106: entries whose
prediction is 0 are masked) and `use_cvmask=True` (sparse_metrics.py:86: entries outside `mvobj_mask > .5` are masked) - only add
entries to the mask, and the reduction masks every entry whose target is 0: they run the same launch on a copy of the target that
is zeroed there (two element-wise device ops, no host synchronisation).
"""
import ctypes
import math

import torch

from . import _lib

_CACHE_KEY = "_monorec_amd_metric_sums"


def sparse_metric_sums_device(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    """(B, 8) float64 DEVICE tensor of per-sample sums - one asynchronous launch, no host synchronisation."""
    pred, gt = data_dict["result"], data_dict["target"]
    if not pred.is_cuda:
        raise RuntimeError("monorec_amd.metrics needs result/target on a HIP device; there is no CPU path")
    lib = _lib.load()
    pred = pred.contiguous().float()
    gt = gt.contiguous().float()
    if not pred_all_valid:                                   # utils/util.py:105-106
        gt = torch.where(pred == 0, torch.zeros_like(gt), gt)
    if use_cvmask:                                           # sparse_metrics.py:86 (KeyError without the mask, like the reference)
        if roi is not None:                                  # the reference crops prediction and target but not the mask: its line raises a shape error
            raise RuntimeError("use_cvmask with a roi: the reference compares the cropped target with the uncropped mvobj_mask and fails; not defined")
        gt = torch.where(data_dict["mvobj_mask"].to(gt.device) > .5, gt, torch.zeros_like(gt))
    b, _, h, w = pred.shape
    assert gt.shape == pred.shape
    sums = torch.empty(b, 8, dtype=torch.float64, device=pred.device)
    roi_arr = (ctypes.c_int32 * 4)(*[int(v) for v in roi]) if roi is not None else None
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    _lib.check(lib.mr_sparse_metric_sums_f32(pred.data_ptr(), gt.data_ptr(), b, h, w, roi_arr,
                                             float(max_distance) if max_distance else 0.0, sums.data_ptr(), stream),
               "mr_sparse_metric_sums_f32")
    return sums


def sparse_metric_sums(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    """(B, 8) float64 CPU tensor of per-sample sums; cached on `data_dict` for the (result, target, roi, dist, options) at hand."""
    pred, gt = data_dict["result"], data_dict["target"]
    if not pred.is_cuda:
        raise RuntimeError("monorec_amd.metrics needs result/target on a HIP device; there is no CPU path")
    mv = data_dict["mvobj_mask"] if use_cvmask else None
    key = (pred.data_ptr(), pred._version, gt.data_ptr(), gt._version, None if roi is None else tuple(roi), max_distance,
           bool(pred_all_valid), None if mv is None else (mv.data_ptr(), mv._version))
    cached = data_dict.get(_CACHE_KEY)
    if cached is not None and cached[0] == key:
        return cached[1]
    out = sparse_metric_sums_device(data_dict, roi, max_distance, pred_all_valid, use_cvmask).cpu()
    data_dict[_CACHE_KEY] = (key, out)
    return out


def metrics_from_sums(s):
    """The seven metric values of one batch, in SPARSE_METRICS order, from its (B, 8) sums (CPU float64)."""
    return [_batch_ratio(s, 1), _batch_ratio(s, 2), _per_sample_rms(s, 3), _per_sample_rms(s, 4),
            _batch_ratio(s, 5), _batch_ratio(s, 6), _batch_ratio(s, 7)]


def _batch_ratio(s, col):
    """mask_mean over the whole batch (utils/util.py:110-118): sum over all samples / number of unmasked entries."""
    n = s[:, 0].sum()
    return torch.tensor(float("nan")) if n == 0 else (s[:, col].sum() / n).float()


def _per_sample_rms(s, col):
    """rmse_base / rmse_log_base (sparse_metrics.py:228-239): sqrt of the per-sample masked mean, mean over samples."""
    return torch.sqrt(s[:, col] / s[:, 0]).mean().float()


def abs_rel_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _batch_ratio(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 1)


def sq_rel_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _batch_ratio(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 2)


def rmse_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _per_sample_rms(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 3)


def rmse_log_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _per_sample_rms(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 4)


def a1_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _batch_ratio(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 5)


def a2_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _batch_ratio(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 6)


def a3_sparse_metric(data_dict, roi=None, max_distance=None, pred_all_valid=True, use_cvmask=False):
    return _batch_ratio(sparse_metric_sums(data_dict, roi, max_distance, pred_all_valid, use_cvmask), 7)


def _variants():
    """The `*_sparse_onlyvalid_metric` (pred_all_valid=False) and `*_sparse_onlydynamic_metric` (use_cvmask=True) wrappers of the reference
    (sparse_metrics.py:158-212), one pair per base metric."""
    g = globals()
    for base in ("a1", "a2", "a3", "rmse", "rmse_log", "abs_rel", "sq_rel"):
        fn = g[f"{base}_sparse_metric"]
        g[f"{base}_sparse_onlyvalid_metric"] = (lambda f: lambda data_dict, roi=None, max_distance=None: f(data_dict, roi, max_distance, False))(fn)
        g[f"{base}_sparse_onlydynamic_metric"] = (lambda f: lambda data_dict, roi=None, max_distance=None: f(data_dict, roi, max_distance, use_cvmask=True))(fn)


_variants()

SPARSE_METRICS = ("abs_rel_sparse_metric", "sq_rel_sparse_metric", "rmse_sparse_metric", "rmse_log_sparse_metric",
                  "a1_sparse_metric", "a2_sparse_metric", "a3_sparse_metric")     # configs/evaluate/eval_monorec.json:53-61
