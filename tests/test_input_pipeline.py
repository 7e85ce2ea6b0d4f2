"""Input pipeline.  CPU: the Pillow-resampling restatement against Pillow's committed outputs, the C host
function for the coefficient tables against the restatement, crop box / intrinsics logic.  GPU: mr_preprocess_image_u8_f32
(behind monorec_amd.input_pipeline) bit for bit against both."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from monorec_amd import _lib, input_pipeline, synth
from oracle import input_oracle

Z = np.load(os.path.join(GOLDEN, "preprocess_cases.npz"))
NAMES = sorted(k for k in Z.files if not k.endswith(".cfg"))
DEV = "cuda:0"


def _case(name):
    h, w, c, oh, ow = (int(v) for v in Z[name + ".cfg"])
    img = synth.make_u8_image(h, w, c, seed=11)
    box = input_oracle.crop_box_for(h, w, oh, ow)
    return img, box, oh, ow, Z[name]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_resize_matches_pillow_fixture(name):
    img, box, oh, ow, want = _case(name)
    x0, y0, x1, y1 = (int(round(v)) for v in box)
    got = input_oracle.resize_bilinear_u8(np.ascontiguousarray(img[y0:y1, x0:x1]), oh, ow)
    assert np.array_equal(got, want)
    try:                                            # and against the Pillow of this machine, when there is one
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.array(Image.fromarray(img).crop(box).resize((ow, oh), resample=Image.BILINEAR)), want)


@pytest.mark.parametrize("sizes", [(740, 512), (370, 256), (60, 96), (33, 33), (47, 20), (7, 3), (5, 64)])
def test_host_coefficient_tables_match_the_restatement(hip_lib, sizes):
    n_in, n_out = sizes
    ks, bounds, coeffs = input_pipeline._axis_tables(hip_lib, n_in, n_out)
    ks_o, b_o, c_o = input_oracle.resample_coeffs(n_in, 0, n_in, n_out)
    assert ks == ks_o and np.array_equal(bounds, b_o) and np.array_equal(coeffs, c_o)
    assert (coeffs.sum(1) - (1 << 22)).__abs__().max() <= ks          # weights sum to one in 22-bit fixed point


def test_crop_box_and_intrinsics_of_the_example_sequence():
    """KITTI sequence 07 (370x1226 -> 256x512): values of SURVEY 3.1 / the reference dataset (pinned in make_golden)."""
    p = np.array([[707.0912, 0, 601.8873, 46.88783], [0, 707.0912, 183.1104, 0.1178601], [0, 0, 1, 0.006203223]])
    intr, box = input_pipeline.compute_target_intrinsics(p, (370, 1226), (256, 512))
    assert tuple(box) == (243.0, 0, 983.0, 370) == tuple(input_oracle.crop_box_for(370, 1226, 256, 512))
    k = input_pipeline.format_intrinsics(intr, (256, 512))
    assert abs(float(k[0, 0]) - 489.2307) < 1e-3 and abs(float(k[0, 2]) - 248.3112) < 1e-3 and abs(float(k[1, 2]) - 126.6926) < 1e-3
    # taller-than-target branch
    _, box2 = input_pipeline.compute_target_intrinsics(p, (400, 600), (256, 512))
    assert box2 == (0, 50.0, 600, 350.0)


def test_no_cpu_fallback():
    with pytest.raises(RuntimeError):
        input_pipeline.ImagePreprocessor((370, 1226), (256, 512), device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_preprocess_is_bit_exact(hip_lib, name):
    img, box, oh, ow, want_u8 = _case(name)
    pre = input_pipeline.ImagePreprocessor(img.shape[:2], (oh, ow), crop_box=box, device=DEV)
    got = pre(img).cpu()
    want = input_oracle.preprocess_image(img, box, oh, ow)
    assert got.shape == (3, oh, ow) and torch.equal(got, want)
    w8 = torch.from_numpy(want_u8.astype(np.float32)) / 255 - .5
    assert torch.equal(got, torch.stack((w8, w8, w8)) if w8.dim() == 2 else w8.permute(2, 0, 1))
    # device-resident input and a caller-provided output buffer give the same bits
    out = torch.full((3, oh, ow), float("nan"), device=DEV)
    pre(torch.from_numpy(img).to(DEV), out=out)
    assert torch.equal(out.cpu(), want)


@pytest.mark.gpu
def test_frame_cache_decodes_each_image_once(hip_lib):
    imgs = {i: synth.make_u8_image(370, 1226, 3, seed=100 + i) for i in range(0, 8)}
    box = input_oracle.crop_box_for(370, 1226, 256, 512)
    pre = input_pipeline.ImagePreprocessor((370, 1226), (256, 512), crop_box=box, device=DEV)
    cache = input_pipeline.FrameCache(lambda i: imgs[i], pre, capacity=4)
    for idx in range(1, 7):                                             # sequential keyframes 1..6, sources idx-1, idx+1
        kf, frames, src = cache.sample(idx, frame_count=2)
        assert src == [idx - 1, idx + 1]
        assert torch.equal(kf.cpu(), input_oracle.preprocess_image(imgs[idx], box, 256, 512))
        assert torch.equal(frames[1].cpu(), input_oracle.preprocess_image(imgs[idx + 1], box, 256, 512))
    assert cache.decoded == 8                                           # the reference decodes 18 images for these 6 samples


def _lidar_fixture():
    z = np.load(os.path.join(GOLDEN, "kitti_example_169.npz"))
    h, w = (int(v) for v in z["input.lidar_shape"])
    png = np.zeros(h * w, dtype=np.uint16)
    png[z["input.lidar_idx"]] = z["input.lidar_val"]
    return png.reshape(h, w), torch.from_numpy(z["input.lidar_target"])


def test_oracle_lidar_target_matches_reference_fixture():
    """The example's annotated-lidar PNG (stored sparsely) -> preprocess_depth_annotated_lidar output of the reference."""
    png, want = _lidar_fixture()
    box = input_oracle.crop_box_for(370, 1226, 256, 512)
    got = input_oracle.lidar_inverse_depth(png, box, 256, 512)
    assert torch.equal(got, want) and int((want > 0).sum()) > 30000
    # collisions exist (79 826 returns -> 35 684 cells), so the last-write-wins order is exercised
    assert int((png > 0).sum()) > int((want > 0).sum())


@pytest.mark.gpu
def test_hip_lidar_target_is_bit_exact(hip_lib):
    png, want = _lidar_fixture()
    box = input_oracle.crop_box_for(370, 1226, 256, 512)
    got = input_pipeline.lidar_inverse_depth(png, box, (256, 512), device=DEV)
    assert torch.equal(got.cpu(), want)
    # no crop, other size, synthetic returns incl. value 1 (inverse depth 256) and the 65535 maximum
    rng = np.random.default_rng(3)
    syn = np.zeros((90, 130), dtype=np.uint16)
    idx = rng.choice(90 * 130, 4000, replace=False)
    syn.reshape(-1)[idx] = rng.integers(1, 65536, 4000).astype(np.uint16)
    syn[0, 0], syn[89, 129] = 1, 65535
    assert torch.equal(input_pipeline.lidar_inverse_depth(syn, None, (32, 48), device=DEV).cpu(),
                       input_oracle.lidar_inverse_depth(syn, None, 32, 48))
