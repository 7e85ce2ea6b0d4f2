"""Synthetic KITTI-shaped keyframe batches and deterministic weights.

There is no dataset and no checkpoint offline, so parity tests, `bench.py` and
`smoke()` all run on inputs produced here: smooth random textures (so SSIM is
non-degenerate), KITTI-example intrinsics, keyframe pose = I and source poses that
are +-0.8 m*k translations along the optical axis with 1 cm lateral jitter
(10 Hz @ ~30 km/h).  `hard_pose=True` moves the whole rig to (-78, 0.6, 23) m to
exercise the fp32 cancellation in inverse(pose_src) @ pose_kf that the reference
has at monorec_model.py:171,207.

Everything is generated with CPU `torch.Generator`s so the very same tensors are
reproduced on the GPU box (same image, same torch build).
"""
import math

import torch
import torch.nn.functional as F

# reference example: 1226x370 KITTI seq 07 centre-cropped/resized to 256x512
# (kitti_odometry_dataset.py:318-349) -> fx=fy=489.23, cx=248.31, cy=126.69
_KITTI_K_256x512 = (489.23, 489.23, 248.31, 126.69)


def make_intrinsics(height, width, batch):
    sy, sx = height / 256.0, width / 512.0
    fx, fy, cx, cy = _KITTI_K_256x512
    k = torch.eye(4, dtype=torch.float32)
    k[0, 0], k[1, 1], k[0, 2], k[1, 2] = fx * sx, fy * sy, cx * sx, cy * sy
    return k.unsqueeze(0).repeat(batch, 1, 1).contiguous()


def _texture(gen, batch, height, width):
    low = torch.rand(batch, 3, max(height // 8, 2), max(width // 8, 2), generator=gen)
    img = F.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)
    img = img + 0.1 * torch.rand(batch, 3, height, width, generator=gen)
    return (img.clamp_(0.0, 1.0) - 0.5).contiguous()


def make_batch(batch=1, height=256, width=512, frames=2, seed=1, hard_pose=False, consistent=True):
    """Input dict with the reference contract (kitti_odometry_dataset.py:260-269).

    consistent=True renders the source frames as horizontally shifted copies of the
    keyframe texture (plus noise) so that the cost volume has real minima; False gives
    independent textures.
    """
    gen = torch.Generator().manual_seed(seed)
    keyframe = _texture(gen, batch, height, width)
    kf_pose = torch.eye(4).unsqueeze(0).repeat(batch, 1, 1)
    if hard_pose:
        kf_pose[:, 0, 3], kf_pose[:, 1, 3], kf_pose[:, 2, 3] = -78.0, 0.6, 23.0
        ang = 0.3
        rot = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        kf_pose[:, :3, :3] = rot
    frame_list, pose_list, intr_list = [], [], []
    for i in range(frames):
        k = i // 2 + 1
        sign = -1.0 if i % 2 == 0 else 1.0
        rel = torch.eye(4).unsqueeze(0).repeat(batch, 1, 1)
        rel[:, 2, 3] = sign * 0.8 * k
        rel[:, 0, 3] = 0.01 * (torch.rand(batch, generator=gen) - 0.5) * 2
        rel[:, 1, 3] = 0.01 * (torch.rand(batch, generator=gen) - 0.5) * 2
        pose = kf_pose @ rel
        if consistent:
            shift = int(round(sign * 3 * k))
            img = torch.roll(keyframe, shifts=shift, dims=3) + 0.02 * (torch.rand(batch, 3, height, width, generator=gen) - 0.5)
            img = img.clamp_(-0.5, 0.5).contiguous()
        else:
            img = _texture(gen, batch, height, width)
        frame_list.append(img)
        pose_list.append(pose.contiguous())
        intr_list.append(make_intrinsics(height, width, batch))
    return {
        "keyframe": keyframe,
        "keyframe_pose": kf_pose.contiguous(),
        "keyframe_intrinsics": make_intrinsics(height, width, batch),
        "frames": frame_list,
        "poses": pose_list,
        "intrinsics": intr_list,
    }


def make_depth_pair(batch=2, height=64, width=96, seed=7, valid_fraction=0.18):
    """Synthetic (prediction, lidar-like sparse target) inverse-depth maps for the metric path: the target is
    zero on ~82 % of the pixels (the annotated KITTI depth of the example has 17.6 % valid pixels)
    and reaches below 1/80 so that the max_distance mask (utils/util.py:101-107) is exercised."""
    gen = torch.Generator().manual_seed(seed)
    pred = 0.0025 + (0.33 - 0.0025) * torch.rand(batch, 1, height, width, generator=gen)
    gt = 0.004 + (0.33 - 0.004) * torch.rand(batch, 1, height, width, generator=gen) ** 2
    keep = torch.rand(batch, 1, height, width, generator=gen) < valid_fraction
    return pred.contiguous(), (gt * keep).contiguous()


def make_metric_flag_inputs(batch, height, width, seed):
    """(prediction with exact zeros on ~10 % of the pixels, sparse target, moving-object mask) for the `pred_all_valid=False` /
    `use_cvmask=True` options of the sparse metrics (sparse_metrics.py:81-212) - seeded, reproducible anywhere."""
    pred, gt = make_depth_pair(batch, height, width, seed)
    g = torch.Generator().manual_seed(seed + 100)
    pred = torch.where(torch.rand(pred.shape, generator=g) < 0.1, torch.zeros_like(pred), pred)
    mv = (torch.rand(pred.shape, generator=g) < 0.35).float()
    return pred.contiguous(), gt, mv.contiguous()


def clone_batch(batch, device=None):
    def cv(v):
        if isinstance(v, torch.Tensor):
            return v.clone().to(device) if device is not None else v.clone()
        if isinstance(v, list):
            return [cv(x) for x in v]
        return v
    return {k: cv(v) for k, v in batch.items()}


WEIGHT_FAMILIES = ("he", "harsh")


def seeded_state_dict(template_state_dict, seed=0, family="he"):
    """Deterministic, key-addressed weights for any state dict with the MonoRec key set.

    Independent of module construction order (unlike torch.manual_seed + default init),
    so the reference, the oracle and the HIP model can all be loaded with identical
    numbers without sharing code.  Conv weights ~ U(+-sqrt(6/fan_in)) (He-uniform, keeps
    activations O(1) through ~20 layers so parity tolerances stay meaningful - PyTorch's
    default bound shrinks the signal by sqrt(3) per layer); biases ~ U(+-1/sqrt(fan_in));
    BatchNorm gets non-trivial affine/statistics so that BN folding is actually exercised.

    family="harsh" (the reduced-multiply kernels were only ever checked on well-conditioned weights) keeps the
    layer-average gain of "he" - so the heads stay out of saturation and the 1e-4 bar stays meaningful - but makes every layer
    ill-conditioned the way a trained checkpoint can be:
      * output channels scaled log-uniformly over [1/3, 3] (a 9x range inside one layer; RMS 1),
      * every 8th filter of a layer with >= 3 taps along an axis is NEAR-CANCELLING: alternating-sign taps of 3x the He bound whose
        sum is ~2 % of their magnitude (on the smooth activations of this net the products cancel - the case in which the larger
        Cook-Toom / Winograd transforms lose digits, model/layers.py:289-314),
      * BatchNorm running_var log-uniform down to 1e-3 on every 8th channel (folded scale up to ~30x its neighbours'), gamma rescaled
        so that the RMS folded scale of the layer is unchanged.
    """
    if family not in WEIGHT_FAMILIES:
        raise ValueError(f"unknown weight family {family!r}")
    harsh = family == "harsh"
    out = {}
    keys = sorted(template_state_dict.keys())
    for idx, key in enumerate(keys):
        ref = template_state_dict[key]
        gen = torch.Generator().manual_seed(1000003 * (seed + 1) + idx)
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=ref.dtype)
            continue
        if key.endswith("running_var"):
            v = 0.5 + torch.rand(shape, generator=gen)
            if harsh:
                tiny = torch.exp(math.log(1e-3) + (math.log(0.5) - math.log(1e-3)) * torch.rand(shape, generator=gen))
                sel = (torch.arange(shape[0]) % 8) == 5
                v = torch.where(sel, tiny, v)
            out[key] = v
            continue
        if key.endswith("running_mean"):
            out[key] = 0.4 * (torch.rand(shape, generator=gen) - 0.5)
            continue
        is_bn = ".bn" in key or ".downsample.1." in key
        if is_bn and key.endswith("weight"):
            # < 1 so the eight residual adds of ResNet-18 do not blow the features up
            out[key] = 0.25 + 0.35 * torch.rand(shape, generator=gen)
            continue
        if is_bn and key.endswith("bias"):
            out[key] = 0.4 * (torch.rand(shape, generator=gen) - 0.5)
            continue
        if len(shape) >= 2:
            transposed = "conv2d_t" in key
            if transposed:  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
                fan_in = shape[0] * 4  # k4/s2: every output pixel sees 2x2 taps of each input channel
            else:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
            bound = math.sqrt(6.0 / fan_in)
        else:
            bound = 0.05
        head = ".predictors." in key or ".classifier." in key
        if head:
            bound *= 0.5  # keep tanh / sigmoid heads out of saturation
        w = ((torch.rand(shape, generator=gen) * 2 - 1) * bound).to(torch.float32)
        if harsh and len(shape) == 4 and not head:
            w = _harshen(w, bound, 1 if transposed else 0, gen)
        out[key] = w
    if harsh:      # gamma rescaled so that the RMS of the folded scale gamma / sqrt(var + eps) is what it would be with the "he" variances
        for key in keys:
            if key.endswith("running_var") and key[:-len("running_var")] + "weight" in out:
                gkey = key[:-len("running_var")] + "weight"
                gen = torch.Generator().manual_seed(1000003 * (seed + 1) + keys.index(key))
                v_he = 0.5 + torch.rand(tuple(out[key].shape), generator=gen)
                g = out[gkey].double()
                rms = lambda var: float(torch.sqrt(torch.mean((g / torch.sqrt(var.double() + 1e-5)) ** 2)))
                out[gkey] = (g * (rms(v_he) / rms(out[key]))).float()
    return out


def _harshen(w, bound, out_dim, gen):
    """The ill-conditioned variant of one conv weight (see seeded_state_dict): `out_dim` = dimension of the output channels."""
    w = w.clone()
    cout = w.shape[out_dim]
    kh, kw = w.shape[2], w.shape[3]
    scale = torch.exp((torch.rand(cout, generator=gen) * 2 - 1) * math.log(3.0))
    if max(kh, kw) >= 3:
        sign = (1 - 2 * ((torch.arange(kh).view(kh, 1) + torch.arange(kw).view(1, kw)) % 2)).to(torch.float32)     # checkerboard / alternating
        cin = w.shape[1 - out_dim]
        amp = 3.0 * bound * (0.5 + 0.5 * torch.rand(cin, generator=gen)) * (1 - 2 * (torch.rand(cin, generator=gen) < 0.5).float())   # sign per input channel
        for c in range(3, cout, 8):
            noise = 1.0 + 0.02 * (torch.rand(cin, kh, kw, generator=gen) * 2 - 1)
            f = amp.view(cin, 1, 1) * sign.view(1, kh, kw) * noise
            if out_dim == 0:
                w[c] = f
            else:
                w[:, c] = f
    view = [1, 1, 1, 1]
    view[out_dim] = cout
    w = w * scale.view(view)
    # layer-average gain of the He family: RMS over the whole tensor back to bound / sqrt(3)
    return (w * ((bound / math.sqrt(3.0)) / float(torch.sqrt(torch.mean(w.double() ** 2))))).to(torch.float32)


def make_pointcloud_case(batch=1, height=64, width=96, seed=3, num_masks=5):
    """Seeded inputs of the point-cloud path (create_pointcloud.py): predicted inverse depth, keyframe, intrinsics,
    pose, `num_masks` cv_mask maps with a few 'moving object' blobs (>= 0.1) and the uniform numbers that stand
    for torch.rand_like in PLYSaver.add_depthmap."""
    gen = torch.Generator().manual_seed(seed)
    inv_depth = 0.0025 + (0.5 - 0.0025) * torch.rand(batch, 1, height, width, generator=gen) ** 2
    image = _texture(gen, batch, height, width)
    intrinsics = make_intrinsics(height, width, batch)
    pose = torch.eye(4).unsqueeze(0).repeat(batch, 1, 1)
    ang = 0.3 * (torch.rand(batch, generator=gen) - 0.5)
    pose[:, 0, 0], pose[:, 0, 2], pose[:, 2, 0], pose[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    pose[:, :3, 3] = 40.0 * (torch.rand(batch, 3, generator=gen) - 0.5)
    cv_masks = []
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    for _ in range(num_masks):
        m = 0.09 * torch.rand(batch, 1, height, width, generator=gen)          # background stays below the 0.1 threshold
        for b in range(batch):
            for _ in range(2):
                cy, cx = torch.rand(2, generator=gen)
                r = 2.0 + 3.0 * float(torch.rand(1, generator=gen))
                blob = ((yy - float(cy) * height) ** 2 + (xx - float(cx) * width) ** 2) < r * r
                m[b, 0][blob] = 0.1 + 0.9 * float(torch.rand(1, generator=gen))
        cv_masks.append(m.contiguous())
    uniform = torch.rand(batch, 1, height, width, generator=gen)
    return dict(inv_depth=inv_depth.contiguous(), image=image, intrinsics=intrinsics, pose=pose.contiguous(),
                cv_masks=cv_masks, uniform=uniform)


def make_u8_image(height, width, channels=3, seed=11):
    """Seeded uint8 test image (smooth texture + noise + hard edges) as numpy (H, W, 3) or (H, W)."""
    import numpy as np
    gen = torch.Generator().manual_seed(seed)
    low = torch.rand(1, channels, max(height // 16, 2), max(width // 16, 2), generator=gen)
    img = F.interpolate(low, size=(height, width), mode="bicubic", align_corners=False)
    img = img + 0.15 * torch.rand(1, channels, height, width, generator=gen)
    img[:, :, height // 3: height // 3 + 5, :] = 1.0                    # a saturated bar and a black bar: clipping paths
    img[:, :, :, width // 4: width // 4 + 3] = 0.0
    a = (img.clamp(0, 1) * 255).round().to(torch.uint8)[0].permute(1, 2, 0).contiguous().numpy()
    return a[:, :, 0].copy() if channels == 1 else a


def make_pixel_depths(batch, depths, height, width, seed=34):
    """Seeded per-pixel depth hypotheses (B, D, H, W) for the `cv_depths` input: the model's uniform inverse-depth ladder,
    smoothly perturbed per pixel by up to +-20 %."""
    gen = torch.Generator().manual_seed(seed)
    base = 1 / torch.linspace(0.0025, 0.33, depths)
    low = torch.rand(batch, depths, max(height // 8, 2), max(width // 8, 2), generator=gen)
    pert = F.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)
    return (base.view(1, depths, 1, 1) * (0.8 + 0.4 * pert)).contiguous()


def make_kitti_tree(root, sequences=(("03", 120, 400), ("07", 200, 300)), frames=16, seed=7, depth_folder="image_depth_annotated"):
    """Write a small KITTI-odometry-shaped directory (the layout the reference's KittiOdometryDataset reads through pykitti):
    sequences/<seq>/{calib.txt, image_0..3/%06d.png, <depth_folder>/%06d.png (16-bit, sparse), mvobj_mask/%06d.npy,
    mask_a.json, mask_b.json}, poses/<seq>.txt and poses_dvso/<seq>.txt.  Seeded; used by the fixtures and the GPU tests."""
    import json
    import os
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(seed)
    for si, (seq, h, w) in enumerate(sequences):
        sdir = os.path.join(str(root), "sequences", seq)
        fx = 0.6 * w
        cx, cy = 0.49 * w + 0.3, 0.47 * h + 0.2
        shifts = (0.0, -0.5372 * fx, 0.0663 * fx, -0.4716 * fx)
        os.makedirs(sdir, exist_ok=True)
        with open(os.path.join(sdir, "calib.txt"), "w") as f:
            for cam, tx in enumerate(shifts):
                p = [fx, 0.0, cx, tx, 0.0, fx, cy, 0.0, 0.0, 0.0, 1.0, 0.0]
                f.write(f"P{cam}: " + " ".join(f"{v:.12e}" for v in p) + "\n")
            f.write("Tr: " + " ".join(f"{v:.12e}" for v in np.eye(4)[:3].reshape(-1)) + "\n")
        for cam in range(4):
            os.makedirs(os.path.join(sdir, f"image_{cam}"), exist_ok=True)
            for i in range(frames):
                img = make_u8_image(h, w, 1 if cam < 2 else 3, seed=seed * 1000 + si * 100 + cam * 30 + i)
                Image.fromarray(img).save(os.path.join(sdir, f"image_{cam}", f"{i:06d}.png"))
        os.makedirs(os.path.join(sdir, depth_folder), exist_ok=True)
        os.makedirs(os.path.join(sdir, "mvobj_mask"), exist_ok=True)
        for i in range(frames):
            d = np.zeros((h, w), dtype=np.uint16)
            hit = rng.rand(h, w) < 0.06
            d[hit] = rng.randint(3 * 256, 80 * 256, size=int(hit.sum())).astype(np.uint16)
            Image.fromarray(d).save(os.path.join(sdir, depth_folder, f"{i:06d}.png"))
            np.save(os.path.join(sdir, "mvobj_mask", f"{i:06d}.npy"), (rng.rand(64, 128) < 0.1).astype(np.float32))
        with open(os.path.join(sdir, "mask_a.json"), "w") as f:
            json.dump({str(i): bool(i % 3) for i in range(frames)}, f)
        with open(os.path.join(sdir, "mask_b.json"), "w") as f:
            json.dump({str(i): bool(i != 7) for i in range(frames - 1)}, f)          # the last index is not listed at all
        for folder, scale in (("poses", 1.0), ("poses_dvso", 0.9)):
            os.makedirs(os.path.join(str(root), folder), exist_ok=True)
            with open(os.path.join(str(root), folder, seq + ".txt"), "w") as f:
                for i in range(frames):
                    a = 0.01 * i * scale
                    rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
                    t = np.array([0.02 * i, -0.01 * i * scale, 0.8 * i * scale + 10.0 * si])
                    f.write(" ".join(f"{v:.12e}" for v in np.hstack([rot, t[:, None]]).reshape(-1)) + "\n")
    return str(root)


# option matrix of the KITTI sample assembly shared by oracle/make_golden.py (reference == oracle) and the GPU tests (product == oracle)
KITTI_OPTION_CASES = {
    "eval_config": dict(frame_count=2, lidar_depth=True, dso_depth=False, use_dso_poses=True),         # configs/evaluate/eval_monorec.json
    "example_defaults": dict(frame_count=2, lidar_depth=True, use_dso_poses=True, use_index_mask=None),  # example/test_monorec.py:18-20
    "dso_only": dict(frame_count=3, dilation=2, offset_d=1, max_length=4),
    "masked_grey_stereo": dict(frame_count=4, use_color=False, lidar_depth=True, dso_depth=False, return_stereo=True,
                               use_index_mask=("mask_a", "mask_b")),
    "mvobj": dict(frame_count=2, lidar_depth=True, return_mvobj_mask=1, return_stereo=True),
}

