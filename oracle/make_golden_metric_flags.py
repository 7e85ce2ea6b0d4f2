"""TEST INFRASTRUCTURE ONLY.  Fixture of the reference's sparse metrics with the two options no evaluation config sets:
pred_all_valid=False (`*_sparse_onlyvalid_metric`, utils/util.py:105-106) and use_cvmask=True (`*_sparse_onlydynamic_metric`,
model/metric_functions/sparse_metrics.py:86).  Runs only in the build container (needs /root/reference):

    python oracle/make_golden_metric_flags.py  ->  tests/golden/sparse_metrics_flags.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from monorec_amd import synth  # noqa: E402
from oracle import ref_shims  # noqa: E402

BASES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def main():
    ref_shims.reference_model_class()                        # installs the import shims
    import model.metric_functions.sparse_metrics as ref      # noqa: the real reference module
    out = {}
    for name, (b, h, w, seed, roi, maxd) in {"flags_eval": (2, 64, 96, 17, None, 80), "flags_roi": (3, 40, 72, 18, [4, 36, 8, 64], None)}.items():
        pred, gt, mv = synth.make_metric_flag_inputs(b, h, w, seed)
        vals = {}
        for base in BASES:
            d = {"result": pred.clone(), "target": gt.clone(), "mvobj_mask": mv.clone()}
            vals[f"{base}_sparse_onlyvalid_metric"] = float(getattr(ref, f"{base}_sparse_onlyvalid_metric")(d, roi, maxd))
            if roi is None:      # (with a roi the reference crops prediction and target but not `mvobj_mask`: its use_cvmask line raises a shape error)
                vals[f"{base}_sparse_onlydynamic_metric"] = float(getattr(ref, f"{base}_sparse_onlydynamic_metric")(d, roi, maxd))
                vals[f"{base}_both"] = float(getattr(ref, f"{base}_sparse_metric")(d, roi, maxd, False, True))
        out[name] = {"config": [b, h, w, seed, roi, maxd], "metrics": vals}
        print(name, "ok:", {k: round(v, 4) for k, v in list(vals.items())[:4]})
    with open(os.path.join(ROOT, "tests", "golden", "sparse_metrics_flags.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
