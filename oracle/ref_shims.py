"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Import shims that let the *unmodified* reference (`/root/reference`, Brummi/MonoRec)
be imported in this container, where `torchvision`, `kornia`, `pykitti`, `skimage`,
`cv2` and `tensorboard` are not installed.

Only used by `oracle/make_golden.py` (runs here, where /root/reference exists) to
 (1) pin `oracle/monorec_oracle.py` against the real reference, and
 (2) generate the committed fixtures under `tests/golden/`.
Nothing here is available on the GPU box and nothing in `tests -m gpu`, `bench.py`
or `smoke()` imports it.

The only arithmetic in this file is the ResNet-18 *topology* that the reference
obtains from `torchvision.models.resnet18` (monorec_model.py:104,113). torchvision is
an un-vendored third-party dependency (environment.yml pins pytorch=1.5.0, torchvision
unpinned -> 0.6.x); its ResNet-18 is the published He et al. architecture:
conv7x7/2(3->64, no bias) + BN + ReLU + maxpool3x3/2 pad1, then 4 stages of 2 BasicBlocks
(conv3x3-BN-ReLU-conv3x3-BN + identity/1x1-stride-2-conv-BN, ReLU) with widths
64/128/256/512, first block of stages 2-4 stride 2. State-dict key names follow
torchvision so that reference checkpoints/keys line up.
"""
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths = [64, 128, 256, 512]
        cin = 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(_BasicBlock(cin, w, stride), _BasicBlock(w, w, 1)))
            cin = w
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def _resnet18(pretrained=False, **_):
    # no network: pretrained weights cannot be fetched; caller loads a state dict afterwards
    return _ResNet18()


class _KittiOdometryStandIn:
    """Just enough of `pykitti.odometry` (un-vendored pip dependency of the reference) for
    data_loader/kitti_odometry_dataset.py to read the example sequence: calibration (P_rect_00/20 and baselines
    from calib.txt), camera file lists / PIL readers and the pose file of `pose_path` (KITTI format: one
    row-major 3x4 cam0->world matrix per line)."""

    def __init__(self, base_path, sequence, **_):
        import os
        from collections import namedtuple
        import numpy as np
        self.sequence = sequence
        self.sequence_path = os.path.join(str(base_path), "sequences", sequence)
        self.pose_path = os.path.join(str(base_path), "poses")
        rows = {}
        with open(os.path.join(self.sequence_path, "calib.txt")) as f:
            for line in f:
                if ":" in line:
                    k, v = line.split(":", 1)
                    rows[k.strip()] = np.array([float(x) for x in v.split()])
        p0, p1, p2, p3 = (rows[f"P{i}"].reshape(3, 4) for i in range(4))
        calib = namedtuple("calib", "P_rect_00 P_rect_10 P_rect_20 P_rect_30 b_gray b_rgb")
        self.calib = calib(p0, p1, p2, p3, -p1[0, 3] / p1[0, 0], (p2[0, 3] - p3[0, 3]) / p2[0, 0])
        for cam, folder in ((0, "image_0"), (1, "image_1"), (2, "image_2"), (3, "image_3")):
            d = os.path.join(self.sequence_path, folder)
            files = sorted(os.path.join(d, n) for n in os.listdir(d)) if os.path.isdir(d) else []
            setattr(self, f"cam{cam}_files", files)
        self.poses = []
        self._load_poses()

    def _reader(self, cam):
        from PIL import Image
        return lambda idx: Image.open(getattr(self, f"cam{cam}_files")[idx])

    def get_cam0(self, idx): return self._reader(0)(idx)
    def get_cam1(self, idx): return self._reader(1)(idx)
    def get_cam2(self, idx): return self._reader(2)(idx)
    def get_cam3(self, idx): return self._reader(3)(idx)

    @property
    def cam0(self): return (self.get_cam0(i) for i in range(len(self.cam0_files)))

    @property
    def cam2(self): return (self.get_cam2(i) for i in range(len(self.cam2_files)))

    def _load_poses(self):
        import os
        import numpy as np
        path = os.path.join(str(self.pose_path), self.sequence + ".txt")
        self.poses = []
        if os.path.exists(path):
            for line in open(path):
                v = np.array([float(x) for x in line.split()])
                if v.size == 12:
                    self.poses.append(np.vstack([v.reshape(3, 4), [0, 0, 0, 1]]))


def _unavailable(name):
    def f(*a, **k):
        raise RuntimeError(f"{name} is a shim placeholder (training/augmentation only)")
    return f


def install():
    """Register shim modules and put the reference on sys.path. Idempotent."""
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.resnet18 = _resnet18
        for n in ("resnet34", "resnet50", "resnet101", "resnet152"):
            setattr(tvm, n, _unavailable(n))
        tvt = types.ModuleType("torchvision.transforms")
        tvt.ColorJitter = type("ColorJitter", (nn.Module,), {})
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = _unavailable("make_grid")
        tv.models, tv.transforms, tv.utils = tvm, tvt, tvu
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm,
                            "torchvision.transforms": tvt, "torchvision.utils": tvu})
    if "kornia" not in sys.modules:
        k = types.ModuleType("kornia")
        ka = types.ModuleType("kornia.augmentation")
        ka.RandomResizedCrop = _unavailable("RandomResizedCrop")
        kg = types.ModuleType("kornia.geometry")
        kgc = types.ModuleType("kornia.geometry.camera")
        kgc.pixel2cam = _unavailable("pixel2cam")
        kgd = types.ModuleType("kornia.geometry.depth")
        kgd.DepthWarper = _unavailable("DepthWarper")
        k.augmentation, k.geometry = ka, kg
        kg.camera, kg.depth = kgc, kgd
        sys.modules.update({"kornia": k, "kornia.augmentation": ka, "kornia.geometry": kg,
                            "kornia.geometry.camera": kgc, "kornia.geometry.depth": kgd})
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "pykitti" not in sys.modules:
        pk = types.ModuleType("pykitti")
        pk.odometry = _KittiOdometryStandIn
        sys.modules["pykitti"] = pk
    import numpy as _np
    for alias, typ in (("float", float), ("int", int)):     # removed numpy aliases the 2020 dataset code still uses
        if not hasattr(_np, alias):
            setattr(_np, alias, typ)
    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        skt = types.ModuleType("skimage.transform")
        skt.resize = _unavailable("skimage.transform.resize")
        sk.transform = skt
        sys.modules.update({"skimage": sk, "skimage.transform": skt})
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_model_class():
    install()
    from model.monorec.monorec_model import MonoRecModel  # noqa: the real reference class
    return MonoRecModel
