"""Run the reference's scripts UNCHANGED on the MI355X path (SURVEY section 8 row b: "scripts untouched").

    cd /path/to/MonoRec
    python -m monorec_amd.dropin evaluate.py --config configs/evaluate/eval_monorec.json
    python -m monorec_amd.dropin create_pointcloud.py --config configs/test/pointcloud_monorec.json

The reference resolves its classes by name at run time - `getattr(model.model, "MonoRecModel")` (evaluate.py:29-31,
create_pointcloud.py:35, utils/parse_config.py:72-89), `getattr(model.metric, name)` (evaluate.py:24),
`from utils import PLYSaver` (create_pointcloud.py:11).  `install()` imports those reference modules (the current
directory / sys.path must be the reference checkout) and rebinds exactly these names:

    model.model.MonoRecModel, model.monorec.monorec_model.MonoRecModel   -> monorec_amd.MonoRecModel
    model.metric.<the seven sparse metrics of eval_monorec.json:53-61>    -> monorec_amd.metrics.<same name>
    utils.PLYSaver, utils.ply_utils.PLYSaver                              -> monorec_amd.pointcloud.PLYSaver

Nothing else of the reference is touched (data loaders, config parser, Evaluater stay the reference's own code) - unless asked:

    python -m monorec_amd.dropin --device-loader evaluate.py --config configs/evaluate/eval_monorec.json

additionally rebinds `data_loader.data_loaders.KittiOdometryDataloader` (looked up by `config.initialize('data_loader',
module_data)`, evaluate.py:20) to `monorec_amd.kitti.KittiOdometryDataloader`: samples are assembled on the device (threaded PNG
decode, one resize launch per new image) instead of by 8 worker processes on the CPU.
"""
import importlib
import os
import runpy
import sys

REBOUND = []        # (module name, attribute) pairs rebound by install(), for inspection / tests


def _rebind(module_name, attr, value):
    try:
        mod = importlib.import_module(module_name)
    except ImportError:
        return False
    setattr(mod, attr, value)
    REBOUND.append((module_name, attr))
    return True


def install(model=True, metrics=True, pointcloud=True, data_loader=False):
    """Rebind the reference's lookup names to the MI355X implementations.  Returns the list of (module, name) rebound."""
    del REBOUND[:]
    from . import MonoRecModel
    if model:
        _rebind("model.monorec.monorec_model", "MonoRecModel", MonoRecModel)
        if not _rebind("model.model", "MonoRecModel", MonoRecModel):
            raise ImportError("monorec_amd.dropin: `model.model` of the reference is not importable - run from the MonoRec "
                              "checkout (or put it on sys.path)")
    if metrics:
        from . import metrics as hip_metrics
        for name in hip_metrics.SPARSE_METRICS:
            _rebind("model.metric", name, getattr(hip_metrics, name))
    if pointcloud:
        from .pointcloud import PLYSaver
        _rebind("utils.ply_utils", "PLYSaver", PLYSaver)
        _rebind("utils", "PLYSaver", PLYSaver)
    if data_loader:
        from .kitti import KittiOdometryDataloader
        if not _rebind("data_loader.data_loaders", "KittiOdometryDataloader", KittiOdometryDataloader):
            raise ImportError("monorec_amd.dropin: `data_loader.data_loaders` of the reference is not importable")
        # (create_pointcloud.py wraps `KittiOdometryDataset` in a torch DataLoader with 8 worker *processes*, :32 - device
        #  samples cannot cross that boundary, so the dataset class itself is left alone)
    return list(REBOUND)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    device_loader = bool(argv) and argv[0] == "--device-loader"
    if device_loader:
        argv = argv[1:]
    if not argv:
        raise SystemExit("usage: python -m monorec_amd.dropin [--device-loader] <reference script.py> [its arguments]")
    script = argv[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or os.getcwd())
    install(data_loader=device_loader)
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
