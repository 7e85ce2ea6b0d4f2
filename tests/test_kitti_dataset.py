"""KITTI sample assembly (data_loader/kitti_odometry_dataset.py:16-311).

CPU: oracle/kitti_oracle.py against the values the unmodified reference class produced on the same synthetic tree
(tests/golden/kitti_tree.json, written by oracle/make_golden.py); host bookkeeping, collation and sharding of
monorec_amd.kitti.  GPU: monorec_amd.kitti.KittiOdometryDataset sample by sample, bit for bit, against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from monorec_amd import kitti, synth
from oracle import input_oracle
from oracle.kitti_oracle import OracleKitti

DEV = "cuda:0"
COMMON = dict(sequences=["03", "07"], depth_folder="image_depth_annotated", target_image_size=(64, 128))
FIXTURE = json.load(open(os.path.join(GOLDEN, "kitti_tree.json")))


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    return synth.make_kitti_tree(tmp_path_factory.mktemp("kitti"))


def _same_tree_as_fixture(tree):
    first = open(os.path.join(tree, "sequences", "03", "image_2", "000000.png"), "rb").read()
    return hashlib.sha1(first).hexdigest() == FIXTURE["tree_sha1"]


@pytest.mark.parametrize("case", sorted(synth.KITTI_OPTION_CASES))
def test_oracle_dataset_matches_reference_fixture(tree, case):
    if not _same_tree_as_fixture(tree):
        pytest.skip("this machine renders the synthetic images differently from the one that wrote the fixture")
    ds = OracleKitti(tree, **dict(COMMON, **synth.KITTI_OPTION_CASES[case]))
    want = FIXTURE["cases"][case]
    assert len(ds) == want["length"]
    for i, (kf_sum, t_sum, t_nz, image_id) in enumerate(want["samples"]):
        data, target = ds[i]
        assert int(data["image_id"]) == image_id and int((target != 0).sum()) == t_nz
        assert float(data["keyframe"].double().sum()) == kf_sum and float(target.double().sum()) == t_sum


@pytest.mark.parametrize("case", sorted(synth.KITTI_OPTION_CASES))
def test_host_bookkeeping_matches_the_oracle(tree, case):
    kw = dict(COMMON, **synth.KITTI_OPTION_CASES[case])
    ds, orc = kitti.KittiOdometryDataset(tree, device="cpu", **kw), OracleKitti(tree, **kw)
    assert len(ds) == len(orc) == FIXTURE["cases"][case]["length"]
    assert ds._dataset_sizes == orc.sizes and [tuple(b) for b in ds._crop_boxes] == [tuple(b) for b in orc.boxes]
    assert all(torch.equal(a, b) for a, b in zip(ds._intrinsics, orc.K))
    if kw.get("use_index_mask", ()) is not None:
        assert ds._indices == orc.indices
    assert ds.get_dataset_index(len(ds)) == (None, None) and ds.get_index(7, 2) == ds._dataset_sizes[0] + 2
    with pytest.raises(IndexError):
        ds[len(ds)]


def test_unsupported_dataset_options_raise(tree):
    for kw in (dict(use_color_augmentation=True), dict(lidar_depth=False, dso_depth=False), dict(lidar_depth=True, annotated_lidar=False)):
        with pytest.raises(NotImplementedError):
            kitti.KittiOdometryDataset(tree, **dict(COMMON, **kw))
    ds = kitti.KittiOdometryDataset(tree, device="cpu", lidar_depth=True, dso_depth=False, **COMMON)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ds[0]


def test_loader_contract_of_the_eval_config(tree):
    """KittiOdometryDataloader(**configs/evaluate/eval_monorec.json:26-50 style args): what evaluate.py / Evaluater touch."""
    import json as js
    args = dict(dataset_dir=tree, depth_folder="image_depth_annotated", batch_size=2, frame_count=2, shuffle=False, validation_split=0,
                num_workers=8, sequences=["03", "07"], target_image_size=[64, 128], use_color=True, use_color_augmentation=False,
                use_dso_poses=True, lidar_depth=True, dso_depth=False, return_stereo=False, device="cpu")
    loader = kitti.KittiOdometryDataloader(**args)
    assert loader.batch_size == 2 and loader.n_samples == len(loader.dataset) == 12 and len(loader) == 6
    assert loader.dataset._decode_workers == 8
    public = {k: v for k, v in loader.dataset.__dict__.items() if not k.startswith("_")}      # evaluate.py:45-52
    js.dumps({k: (list(v) if isinstance(v, np.ndarray) else v) for k, v in public.items()})
    assert public["target_image_size"] == (64, 128) and public["length"] == 12
    with pytest.raises(NotImplementedError):
        kitti.KittiOdometryDataloader(**dict(args, shuffle=True))


def test_collate_and_batch_sharding():
    class Fake:
        def __len__(self):
            return 7

        def __getitem__(self, i):
            t = torch.full((3, 2, 2), float(i))
            return {"keyframe": t, "frames": [t + 1, t + 2], "sequence": torch.tensor([i], dtype=torch.int32)}, t[:1]

    batches = list(kitti.DeviceLoader(Fake(), batch_size=2))
    assert [b[1].shape[0] for b in batches] == [2, 2, 2, 1] and len(kitti.DeviceLoader(Fake(), batch_size=2)) == 4
    data, target = batches[1]
    assert data["keyframe"].shape == (2, 3, 2, 2) and data["sequence"].shape == (2, 1) and target.shape == (2, 1, 2, 2)
    assert isinstance(data["frames"], list) and data["frames"][1].shape == (2, 3, 2, 2) and float(data["frames"][1][1, 0, 0, 0]) == 5.0
    shards = [[int(b[0]["sequence"][0]) for b in kitti.DeviceLoader(Fake(), batch_size=2, rank=r, world_size=2)] for r in range(2)]
    assert shards == [[0, 4], [2, 6]]                                  # whole batches, round-robin


# ---------------------------------------------------------------------------------------------- GPU
def _equal(a, b, what):
    assert a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape), (what, a.dtype, b.dtype, a.shape, b.shape)
    assert torch.equal(a.cpu(), b), what


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(synth.KITTI_OPTION_CASES))
def test_hip_dataset_samples_are_bit_exact(hip_lib, tree, case):
    kw = dict(COMMON, **synth.KITTI_OPTION_CASES[case])
    ds, orc = kitti.KittiOdometryDataset(tree, device=DEV, decode_workers=3, **kw), OracleKitti(tree, **kw)
    assert len(ds) == len(orc) > 0
    for i in range(len(ds)):
        (data, target), (odata, otarget) = ds[i], orc[i]
        assert sorted(data) == sorted(odata)
        for k, want in odata.items():
            if isinstance(want, list):
                assert len(data[k]) == len(want)
                for j, w in enumerate(want):
                    _equal(data[k][j], w, (case, i, k, j))
            else:
                _equal(data[k], want, (case, i, k))
        _equal(target, otarget, (case, i, "target"))
        assert data["keyframe"].is_cuda and target.is_cuda
    decoded = sum(c.decoded for c in ds._caches.values())
    per_cam = sum(len(set(range(16))) for _ in ds._caches)
    assert decoded <= per_cam, "every image is decoded and resized at most once per sweep"     # the reference: 1 + frame_count times
    ds.close()


@pytest.mark.gpu
def test_hip_dso_target_is_bit_exact_at_kitti_size(hip_lib):
    rng = np.random.RandomState(5)
    png = np.zeros((370, 1226), dtype=np.uint16)
    hit = rng.rand(370, 1226) < 0.2
    png[hit] = rng.randint(1, 65535, size=int(hit.sum())).astype(np.uint16)
    from monorec_amd import input_pipeline
    for box, size, par in ((input_oracle.crop_box_for(370, 1226, 256, 512), (256, 512), (370, 1226, 707.0912)),
                           (None, (64, 96), (375, 1242, 718.856)),                    # PNG smaller than the image: coordinates rescale
                           (input_oracle.crop_box_for(375, 1242, 64, 64), (64, 64), (375, 1242, 718.856))):
        got = input_pipeline.dso_inverse_depth(png, par, box, size, device=DEV)
        _equal(got, input_oracle.dso_inverse_depth(png, par, box, *size), (box, size))


@pytest.mark.gpu
def test_eval_config_end_to_end(hip_lib, tree):
    """The evaluation flow of configs/evaluate/eval_monorec.json on the device: KittiOdometryDataloader -> MonoRecModel -> fused
    metrics -> Evaluater bookkeeping, against the same flow fed with the oracle's CPU-assembled batches and the oracle's metrics."""
    import math
    from monorec_amd import MonoRecModel, evaluate
    from oracle import monorec_oracle as orc
    args = dict(COMMON, **synth.KITTI_OPTION_CASES["eval_config"])
    loader = kitti.KittiOdometryDataloader(dataset_dir=tree, batch_size=2, shuffle=False, num_workers=4, device=DEV, **args)
    model = MonoRecModel(cv_depth_steps=8)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).eval()
    log = evaluate.Evaluater(model, roi=None, max_distance=80).eval(loader)
    orc_ds = OracleKitti(tree, **args)
    per_batch = []
    for lo in range(0, len(orc_ds), 2):
        data, target = kitti.collate([orc_ds[i] for i in range(lo, lo + 2)])
        with torch.no_grad():
            res = model(synth.clone_batch(data, DEV))["result"].cpu().clone()
        vals = orc.sparse_metrics(res, target, None, 80)
        per_batch.append([float(vals[k]) for k in evaluate._metrics.SPARSE_METRICS])
    assert log["valid_batches"] == len(per_batch) == 6
    want = np.mean(np.array(per_batch), axis=0)
    for got, w in zip(log["metrics"], want):
        assert math.isclose(got, w, rel_tol=2e-5, abs_tol=1e-7), (log["metrics"], want.tolist())
    loader.dataset.close()
