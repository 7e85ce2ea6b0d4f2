"""Point-cloud path.  CPU: the oracle against the committed outputs of the reference's PLYSaver /
create_pointcloud.py mask lines.  GPU: monorec_amd.pointcloud (mr_static_mask_f32, mr_pointcloud_append_f32) against both."""
import io
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from monorec_amd import synth
from oracle import monorec_oracle as orc

CASES = json.load(open(os.path.join(GOLDEN, "pointcloud_cases.json")))
DEV = "cuda:0"


def _load(name):
    cfg = CASES[name]
    z = np.load(os.path.join(GOLDEN, f"pointcloud_{name}.npz"))
    case = synth.make_pointcloud_case(cfg["b"], cfg["h"], cfg["w"], cfg["seed"])
    return cfg, z, case


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_pointcloud_matches_reference_fixture(name):
    cfg, z, case = _load(name)
    masks = [orc.static_mask(m, 32) for m in case["cv_masks"]]
    assert [float(m.sum()) for m in masks] == list(z["static_mask_sum"])
    assert np.array_equal(np.packbits(masks[0].numpy().astype(np.uint8)), z["static_mask0"])
    rec = orc.pointcloud_records(case["inv_depth"], case["image"], case["intrinsics"], case["pose"], cfg["min_d"], cfg["max_d"],
                                 cfg["roi"], cfg["dropout"], case["uniform"], masks if cfg["use_mask"] else None, 1)
    want = torch.from_numpy(z["records"])
    assert rec.shape == want.shape == (cfg["points"], 6)
    # selection is exact; coordinates may differ in the last bits on another host CPU (sgemm kernels)
    assert torch.allclose(rec, want, rtol=1e-5, atol=1e-4)


def test_static_mask_definition():
    cv = torch.zeros(1, 1, 40, 50)
    cv[0, 0, 20, 25] = 0.1            # exactly the threshold counts as moving
    cv[0, 0, 0, 0] = 0.0999
    m = orc.static_mask(cv, 32)
    assert m[0, 0, 20, 25] == 0 and m[0, 0, 4, 9] == 0 and m[0, 0, 36, 41] == 0      # +-16 window
    assert m[0, 0, 3, 25] == 1 and m[0, 0, 20, 42] == 1 and m[0, 0, 0, 0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_pointcloud_matches_reference_fixture_and_oracle(hip_lib, name):
    from monorec_amd.pointcloud import PLYSaver, static_mask
    cfg, z, case = _load(name)
    dev = lambda t: t.to(DEV)
    masks = [static_mask(dev(m), 32) for m in case["cv_masks"]]
    ref_masks = [orc.static_mask(m, 32) for m in case["cv_masks"]]
    for a, b in zip(masks, ref_masks):
        assert torch.equal(a.cpu(), b)                                   # integer logic: exact
    saver = PLYSaver(cfg["h"], cfg["w"], min_d=cfg["min_d"], max_d=cfg["max_d"], batch_size=cfg["b"], roi=cfg["roi"],
                     dropout=cfg["dropout"], capacity=64)               # tiny capacity: exercises the growth path
    saver.to(DEV)
    for _ in range(2):                                                   # appended twice, one host copy at the end
        saver.add_depthmap(dev(case["inv_depth"]), dev(case["image"]), dev(case["intrinsics"]), dev(case["pose"]),
                           static_masks=masks if cfg["use_mask"] else None, min_hits=1, uniform=dev(case["uniform"]))
    got = torch.tensor(saver.data, dtype=torch.float32).view(-1, 6)
    want = torch.from_numpy(z["records"])
    assert got.shape[0] == 2 * cfg["points"]
    for half in (got[:cfg["points"]], got[cfg["points"]:]):
        assert torch.allclose(half, want, rtol=1e-5, atol=1e-4), float((half - want).abs().max())
        assert torch.equal(half[:, 3:], want[:, 3:])                     # colours: exact
    here = orc.pointcloud_records(case["inv_depth"], case["image"], case["intrinsics"], case["pose"], cfg["min_d"], cfg["max_d"],
                                  cfg["roi"], cfg["dropout"], case["uniform"], ref_masks if cfg["use_mask"] else None, 1)
    assert torch.allclose(got[:cfg["points"]], here, rtol=1e-5, atol=1e-4)
    # .ply bytes: header of utils/ply_utils.py:18-32 + the float records
    buf = io.BytesIO()
    saver.save(buf)
    head, _, body = buf.getvalue().partition(b"end_header\n")
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and f"element vertex {2 * cfg['points']}".encode() in head
    assert np.array_equal(np.frombuffer(body, dtype="<f4"), got.numpy().reshape(-1))


@pytest.mark.gpu
def test_builder_runs_the_reference_loop(hip_lib):
    """create_pointcloud.py:66-102 with 7 keyframes through the HIP model -> 3 depth maps reach the saver; compared with
    the oracle chain (static masks, vote, add_depthmap) on the same model outputs."""
    from monorec_amd import MonoRecModel
    from monorec_amd.pointcloud import PLYSaver, PointcloudBuilder
    model = MonoRecModel(cv_depth_steps=8, hip_in_flight=1)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).eval()
    saver = PLYSaver(128, 192, min_d=3, max_d=400, batch_size=1, roi=[4, 120, 8, 180], dropout=0)
    saver.to(DEV)
    builder = PointcloudBuilder(saver, mask_fill=32, buffer_length=5, min_hits=1)
    kept = []
    with torch.no_grad():
        for i in range(7):
            data = synth.clone_batch(synth.make_batch(1, 128, 192, 2, seed=20 + i), DEV)
            res = dict(model(data))
            # random-init weights call almost every pixel "moving"; keep only the strongest response of each keyframe
            # so that the 33x33 dilation and the 5-frame vote leave something to compare
            cm = res["cv_mask"]
            res["cv_mask"] = (cm >= cm.flatten().topk(1).values[-1]).float()
            kept.append({k: (v.cpu().clone() if torch.is_tensor(v) else v) for k, v in
                         dict(depth=res["result"], cv_mask=res["cv_mask"], keyframe=data["keyframe"],
                              K=data["keyframe_intrinsics"], pose=data["keyframe_pose"]).items()})
            builder.add(data, res)
    got = torch.tensor(saver.data, dtype=torch.float32).view(-1, 6)
    want = []
    for j in range(2, 5):                                                # key index 2 of each full 5-window
        win = kept[j - 2:j + 3]
        masks = [orc.static_mask(e["cv_mask"], 32) for e in win]
        k = kept[j]
        want.append(orc.pointcloud_records(k["depth"], k["keyframe"], k["K"], k["pose"], 3, 400, [4, 120, 8, 180], 0, None, masks, 1))
    want = torch.cat(want)
    assert got.shape == want.shape and got.shape[0] > 1000, (got.shape, want.shape)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_unchanged_reference_loop_body_with_two_keyframes_in_flight(hip_lib):
    """The loop body of the reference's create_pointcloud.py:66-102, statement for statement (buffers of 5, no clones,
    `depth *= mask` in place), over `model(data)` with the default two in-flight slots: `forward()` must hand out tensors the
    caller owns (monorec_model.py:713-727 allocates its outputs), otherwise depth_buffer[key_index] is the slot the newest
    keyframe has just overwritten.  Compared frame for frame with the oracle chain on per-forward copies of the outputs."""
    import torch.nn.functional as F
    from monorec_amd import MonoRecModel
    from monorec_amd.pointcloud import PLYSaver
    model = MonoRecModel(cv_depth_steps=8, hip_in_flight=2)
    assert model._in_flight == 2 and MonoRecModel(cv_depth_steps=8)._in_flight == 4       # (the default: four slots, one stream each, since round 5)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).eval()
    max_d, min_d, mask_fill, use_mask, roi = 400, 3, 32, True, [4, 120, 8, 180]
    plysaver = PLYSaver(128, 192, min_d=min_d, max_d=max_d, batch_size=1, roi=roi, dropout=0)
    plysaver.to(DEV)
    loader = [(synth.make_batch(1, 128, 192, 2, seed=20 + i), None) for i in range(7)]
    kept, per_frame = [], []

    pose_buffer = []
    intrinsics_buffer = []
    mask_buffer = []
    keyframe_buffer = []
    depth_buffer = []

    buffer_length = 5
    min_hits = 1
    key_index = buffer_length // 2

    with torch.no_grad():
        for i, (data, target) in enumerate(loader):
            data = synth.clone_batch(data, DEV)
            result = model(data)
            if not isinstance(result, dict):
                result = {"result": result[0]}
            output = result["result"]
            if "cv_mask" not in result:
                result["cv_mask"] = output.new_zeros(output.shape)
            # (test only) random-init weights call almost every pixel "moving": keep the strongest response per keyframe
            result["cv_mask"] = (result["cv_mask"] >= result["cv_mask"].flatten().topk(1).values[-1]).float()
            kept.append(dict(depth=output.cpu().clone(), cv_mask=result["cv_mask"].cpu().clone(), keyframe=data["keyframe"].cpu().clone(),
                             K=data["keyframe_intrinsics"].cpu().clone(), pose=data["keyframe_pose"].cpu().clone()))
            mask = (result["cv_mask"] >= .1).to(dtype=torch.float32)
            mask = (F.conv2d(mask, mask.new_ones((1, 1, mask_fill + 1, mask_fill + 1)), padding=mask_fill // 2) < 1).to(dtype=torch.float32)

            pose_buffer += data["keyframe_pose"]
            intrinsics_buffer += [data["keyframe_intrinsics"]]
            mask_buffer += [mask]
            keyframe_buffer += [data["keyframe"]]
            depth_buffer += [output]

            if len(pose_buffer) >= buffer_length:
                pose = pose_buffer[key_index]
                intrinsics = intrinsics_buffer[key_index]
                keyframe = keyframe_buffer[key_index]
                depth = depth_buffer[key_index]

                mask = (torch.sum(torch.stack(mask_buffer), dim=0) > buffer_length - min_hits).to(dtype=torch.float32)
                if use_mask:
                    depth *= mask

                before = len(plysaver.data) // 6
                plysaver.add_depthmap(depth, keyframe, intrinsics, pose)
                per_frame.append(torch.tensor(plysaver.data[before * 6:], dtype=torch.float32).view(-1, 6))

                del pose_buffer[0]
                del intrinsics_buffer[0]
                del mask_buffer[0]
                del keyframe_buffer[0]
                del depth_buffer[0]
    assert len(per_frame) == 3
    for n, j in enumerate(range(2, 5)):                                  # key index 2 of each full 5-window
        masks = [orc.static_mask(e["cv_mask"], mask_fill) for e in kept[j - 2:j + 3]]
        k = kept[j]
        want = orc.pointcloud_records(k["depth"], k["keyframe"], k["K"], k["pose"], min_d, max_d, roi, 0, None, masks, min_hits)
        got = per_frame[n]
        assert got.shape == want.shape and got.shape[0] > 300, (n, got.shape, want.shape)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-4), (n, float((got - want).abs().max()))
