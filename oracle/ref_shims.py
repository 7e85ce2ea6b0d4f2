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
    for name in ("pykitti", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
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
