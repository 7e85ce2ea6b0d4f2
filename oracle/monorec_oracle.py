"""TEST INFRASTRUCTURE ONLY - CPU oracle for the MonoRec cost-volume inference path.

A functional, state-dict driven restatement (torch CPU fp32 ops, no nn.Module graph) of
what `MonoRecModel.forward` computes in eval mode with the default inference options
(pretrain_mode=0, use_mono, use_ssim=True, sfcv_mult_mask=True, no augmentation):

    reference  /root/reference/model/monorec/monorec_model.py:672-729  (forward glue)
               :150-280  CostVolumeModule.forward      -> cost_volume()
               :118-129  ResnetEncoder.forward         -> resnet_features()
               :345-385  MaskModule.forward            -> mask_module()
               :526-557  DepthModule.forward           -> depth_module()
               /root/reference/model/layers.py:43-71,119-139,241-252,289-356,380-400

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file, and only as the checker / the timed CPU baseline - never as the product path
(`monorec_amd` never imports `oracle`).

Pinning status: PINNED against the real reference run in the build container
(`oracle/make_golden.py` imports /root/reference through `oracle/ref_shims.py`, asserts this
oracle equals it on every stage, and writes `tests/golden/*.npz`).  The reference itself ships
no golden vectors or tests, so the fixtures are outputs of the reference
code with seeded synthetic weights/inputs - see `oracle/make_golden.py`.

The arithmetic of every stage lives in a third-party dependency of the reference (PyTorch,
pinned `pytorch=1.5.0` in environment.yml; torchvision for the ResNet-18 topology).  The
restatement calls the same ATen operators with the same operand shapes so that operator-level
rounding (MKL/oneDNN accumulation order) is shared with the reference on the same host.
"""
import math

import torch
import torch.nn.functional as F

SSIM_C1 = 0.01 ** 2   # layers.py:116
SSIM_C2 = 0.03 ** 2   # layers.py:117


# ----------------------------------------------------------------------------------------
# cost volume  (monorec_model.py:150-280)
# ----------------------------------------------------------------------------------------
def depth_hypotheses(inv_depth_min, inv_depth_max, steps):
    """monorec_model.py:184 - far (1/inv_depth_max) to near (1/inv_depth_min)."""
    return 1 / torch.linspace(float(inv_depth_max), float(inv_depth_min), int(steps))


def pixel_grid(height, width):
    """layers.py:49-54 - homogeneous pixel coordinates [x; y; 1], x fastest, shape (1,3,HW)."""
    ys, xs = torch.meshgrid(torch.arange(0., float(height)), torch.arange(0., float(width)), indexing="ij")
    ones = torch.ones(1, 1, height * width)
    return torch.cat([torch.stack([xs.reshape(-1), ys.reshape(-1)], 0).unsqueeze(0), ones], 1)


def border_mask(height, width, radius):
    """monorec_model.py:282-284 - ones with a `radius` wide zero frame, shape (1,1,H,W)."""
    m = torch.zeros(1, 1, height, width)
    m[:, :, radius:height - radius, radius:width - radius] = 1
    return m


def projection_matrix(src_intrinsics, src_pose, kf_pose):
    """monorec_model.py:171,207 + layers.py:65 - (K_f @ (inverse(pose_f) @ pose_kf))[:3,:] for one sample.

    All operands 4x4 fp32; returns (1,3,4)."""
    t = torch.inverse(src_pose) @ kf_pose
    return torch.matmul(src_intrinsics.unsqueeze(0), t.unsqueeze(0))[:, :3, :]


def sample_grid(cam_points, proj, height, width):
    """layers.py:65-70 + monorec_model.py:208: project, normalise with the (W-1)/(H-1) convention, clamp."""
    d = cam_points.shape[0]
    pc = torch.matmul(proj, cam_points)
    xy = pc[:, :2, :] / (pc[:, 2:3, :] + 1e-7)
    xy[:, 0, :] /= width - 1
    xy[:, 1, :] /= height - 1
    xy = (xy - 0.5) * 2
    return xy.view(d, 2, height, width).permute(0, 2, 3, 1).clamp(-2, 2)


def ssim_distance(x, y):
    """layers.py:119-137 - (1-SSIM)/2 with 3x3 box means and reflection padding."""
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    mu_x_sq, mu_y_sq, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    sig_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x_sq
    sig_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y_sq
    sig_xy = F.avg_pool2d(x * y, 3, 1) - mu_xy
    num = (2 * mu_xy + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    den = (mu_x_sq + mu_y_sq + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def cost_volume(batch, inv_depth_min=0.33, inv_depth_max=0.0025, steps=32, patch_size=3,
                channel_weights=(5 / 32, 16 / 32, 11 / 32), alpha=10, stages=None, use_ssim=True, cv_depths=None,
                sfcv_mult_mask=True):
    """CostVolumeModule.forward (monorec_model.py:150-280), use_mono, sfcv_mult_mask=True; use_ssim selects the photometric
    term (:227-243): True SSIM distance, False absolute difference, 2 the 0.85/0.15 mix, 3 3x3-averaged absolute difference.

    Returns (cost_volume (B,D,H,W), [single_frame_cv (B,D,H,W)] * F).
    If `stages` is a dict it receives per-sample intermediates (grid, warped, sad, valid, weight)."""
    keyframe = batch["keyframe"]
    frames, intrs, poses = batch["frames"], batch["intrinsics"], batch["poses"]
    b, c, h, w = keyframe.shape
    nf = len(frames)
    radius = patch_size // 2 + 1                                                     # :139
    sad_kernel = (torch.tensor(channel_weights) / (patch_size ** 2)).view(1, c, 1, 1, 1) \
        .repeat(1, 1, 1, patch_size, patch_size)                                     # :141-142
    depths = depth_hypotheses(inv_depth_min, inv_depth_max, steps)                   # :184
    coord = pixel_grid(h, w)
    ones = torch.ones(1, 1, h * w)
    mask0 = border_mask(h, w, radius)
    cvs, sfcvs = [], [[] for _ in range(nf)]
    for n in range(b):                                                               # :193
        inv_k = torch.inverse(batch["keyframe_intrinsics"][n]).unsqueeze(0)          # :198
        rays = inv_k[:, :3, :3] @ coord                                              # :199
        if cv_depths is not None:                                                    # per-pixel hypotheses, :181-182
            pts = cv_depths[n].reshape(steps, 1, -1) * rays
        else:
            pts = depths.view(steps, 1, 1) * rays                                    # :200
        pts = torch.cat([pts, ones.expand(steps, -1, -1)], 1)                        # :201
        warped, valid = [], []
        grids = []
        for f in range(nf):                                                          # :206
            proj = projection_matrix(intrs[f][n], poses[f][n], batch["keyframe_pose"][n])
            grid = sample_grid(pts, proj, h, w)                                      # :208
            grids.append(grid)
            warped.append(F.grid_sample(frames[f][n:n + 1].expand(steps, -1, -1, -1), grid,
                                        mode="bilinear", padding_mode="zeros", align_corners=False))   # :215
            wm = F.grid_sample(mask0.expand(steps, -1, -1, -1), grid,
                               mode="bilinear", padding_mode="zeros", align_corners=False)              # :218
            valid.append(mask0[0] * torch.min(wm != 0, dim=0)[0])                    # :219
        warped = torch.stack(warped, 1)                       # (D,F,C,H,W)           :223
        valid = torch.stack(valid)                            # (F,1,H,W)             :225
        nb = steps * nf
        if not use_ssim:                                                             # :227-228
            diff = torch.abs(warped - keyframe[n])
        elif use_ssim is True or use_ssim == 1:
            diff = ssim_distance(warped.view(nb, c, h, w) + .5,
                                 keyframe[n].unsqueeze(0).expand(nb, -1, -1, -1) + .5)   # :231-232
        elif use_ssim == 2:                                                          # :234-239
            diff = ssim_distance(warped.view(nb, c, h, w) + .5,
                                 keyframe[n].unsqueeze(0).expand(nb, -1, -1, -1) + .5).view(steps, nf, c, h, w)
            diff = 0.85 * diff + 0.15 * torch.abs(warped - keyframe[n])
        else:                                                                        # :240-243
            diff = F.avg_pool2d(torch.abs(warped - keyframe[n]).view(nb, c, h, w), kernel_size=3, stride=1, padding=1)
        diff = diff.view(steps, nf, c, h, w).permute(1, 2, 0, 3, 4)                  # :233,246
        sad = F.conv3d(diff, sad_kernel, padding=(0, patch_size // 2, patch_size // 2)).squeeze(1)   # :247
        if sfcv_mult_mask:
            sfcv = (1 - sad * 2) * valid                                             # :251
        else:                                                                        # :253
            sfcv = (1 - sad * 2) * (torch.any(warped != 0, dim=2) | torch.all(warped == keyframe[n], dim=2)).permute(1, 0, 2, 3)
        for f in range(nf):
            sfcvs[f].append(sfcv[f])
        e = torch.exp(-alpha * torch.pow(sad - torch.min(sad, dim=1, keepdim=True)[0], 2))     # :257
        weight = 1 - 1 / (steps - 1) * (torch.sum(e, dim=1, keepdim=True) - 1)       # :258
        weight = weight * valid                                                      # :260
        cv = torch.sum(sad * weight, dim=0)                                          # :262
        wsum = torch.sum(weight, dim=0).squeeze()                                    # :264
        nz = wsum != 0
        cv[:, nz] /= wsum[nz]                                                        # :266
        cv = 1 - 2 * cv                                                              # :268
        cv[:, ~nz] = 0                                                               # :269
        cvs.append(cv)
        if stages is not None:
            stages.setdefault("grid", []).append(torch.stack(grids))        # (F,D,H,W,2)
            stages.setdefault("warped", []).append(warped)
            stages.setdefault("sad", []).append(sad)
            stages.setdefault("valid", []).append(valid)
            stages.setdefault("weight", []).append(weight)
    return torch.stack(cvs), [torch.stack(s) for s in sfcvs]


# ----------------------------------------------------------------------------------------
# conv building blocks (layers.py)
# ----------------------------------------------------------------------------------------
def same_pad(n, k, s):
    """layers.py:249-251 - TF 'same': total s*(ceil(n/s)-1)+k-n, floor on the low side."""
    total = s * (math.ceil(n / s) - 1) + k - n
    return math.floor(total / 2), math.ceil(total / 2)


def conv_same(x, w, b, stride=(1, 1)):
    """PadSameConv2d + Conv2d (layers.py:241-252, 329)."""
    kh, kw = w.shape[2], w.shape[3]
    pt, pb = same_pad(x.shape[2], kh, stride[0])
    pl, pr = same_pad(x.shape[3], kw, stride[1])
    return F.conv2d(F.pad(x, [pl, pr, pt, pb]), w, b, stride=stride)


def lrelu(x):
    return F.leaky_relu(x, 0.1)


def conv_relu(sd, prefix, x):
    """layers.ConvReLU (layers.py:332-335)."""
    return lrelu(conv_same(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"]))


def conv_relu2(sd, prefix, x, stride=1):
    """layers.ConvReLU2 (layers.py:308-314): k x 1 (stride (s,1)) then 1 x k (stride (1,s))."""
    t = lrelu(conv_same(x, sd[prefix + ".conv_y.weight"], sd[prefix + ".conv_y.bias"], (stride, 1)))
    return lrelu(conv_same(t, sd[prefix + ".conv_x.weight"], sd[prefix + ".conv_x.bias"], (1, stride)))


def upconv(sd, prefix, x):
    """layers.Upconv (layers.py:353-356): nearest x2, pad (0,1,0,1), conv 2x2, no activation."""
    t = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv_same(t, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"])


def refine(sd, prefix, x):
    """layers.Refine (layers.py:393-400): ConvTranspose2d(k4,s2) -> LeakyReLU -> centre crop to 2x."""
    t = lrelu(F.conv_transpose2d(x, sd[prefix + ".conv2d_t.weight"], sd[prefix + ".conv2d_t.bias"], stride=2))
    return t[:, :, 1:-1, 1:-1]


# ----------------------------------------------------------------------------------------
# ResNet-18 encoder  (monorec_model.py:118-129; topology from torchvision, see ref_shims.py)
# ----------------------------------------------------------------------------------------
def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, 1e-5)


def _basic_block(sd, prefix, x, stride):
    y = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, prefix + ".bn2", F.conv2d(y, sd[prefix + ".conv2.weight"], None, 1, 1))
    if (prefix + ".downsample.0.weight") in sd:
        x = _bn(sd, prefix + ".downsample.1", F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0))
    return F.relu(y + x)


def resnet_features(sd, image, prefix="_feature_extractor.encoder"):
    """ResnetEncoder.forward (monorec_model.py:118-129). `image` is keyframe + 0.5."""
    x = (image - 0.45) / 0.225
    x = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"], None, 2, 3)))
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            x = _basic_block(sd, f"{prefix}.layer{li}.{bi}", x, 2 if (li > 1 and bi == 0) else 1)
        feats.append(x)
    return feats


# ----------------------------------------------------------------------------------------
# MaskModule (monorec_model.py:345-385)
# ----------------------------------------------------------------------------------------
def mask_module(sd, sfcvs, feats, prefix="att_module", use_cv=True, use_features=True, simple_input=None):
    """MaskModule.forward (:345-385); with `simple_input` = (keyframe, previous inverse depth) SimpleMaskModule.forward
    (:444-473): one encoder pass over cat(non-zero mean of the single-frame volumes, keyframe, previous depth), same decoder."""
    cv_feats = []
    if not use_cv:
        sfcvs = [c * 0 for c in sfcvs]                                              # :352-353
    if not use_features:
        feats = [f * 0 for f in feats]                                              # :354-355
    if simple_input is not None:
        stacked = torch.stack(sfcvs, dim=0)                                         # :447-449
        mean = stacked.sum(dim=0) / (stacked != 0).to(dtype=torch.float32).sum(dim=0).clamp_min(1)
        sfcvs = [torch.cat([mean, simple_input[0], simple_input[1]], dim=1)]         # :453
    for cv in sfcvs:                                                                # :357
        x = cv
        for i in range(5):
            if i > 0:
                x = F.max_pool2d(x, 2)
            a, b2 = (0, 1) if i == 0 else (1, 2)          # index 0 of stages 1-4 is the MaxPool
            x = conv_relu(sd, f"{prefix}.enc.{i}.{a}", x)
            x = conv_relu(sd, f"{prefix}.enc.{i}.{b2}", x)
            if len(cv_feats) == i:
                cv_feats.append(x)
            else:
                cv_feats[i] = torch.max(cv_feats[i], x)                            # :365
    x = torch.cat([cv_feats[4], feats[3]], 1)                                       # :372
    for i in range(4):
        x = upconv(sd, f"{prefix}.dec.{i}.0", x)
        if i == 3:
            x = torch.cat([cv_feats[3 - i], x], 1)                                  # :377
        else:
            x = torch.cat([cv_feats[3 - i], feats[2 - i], x], 1)                    # :374,380
        x = conv_relu(sd, f"{prefix}.dec.{i}.1", x)
        x = conv_relu(sd, f"{prefix}.dec.{i}.2", x)
    return torch.sigmoid(F.conv2d(x, sd[prefix + ".classifier.0.weight"], sd[prefix + ".classifier.0.bias"]))


# ----------------------------------------------------------------------------------------
# DepthModule (monorec_model.py:526-557)
# ----------------------------------------------------------------------------------------
_DEPTH_ENC = ((7, 1), (7, 2), (5, 2), (5, 2), (3, 2))     # (kernel, stride) of the first ConvReLU2 per stage :487-500


def depth_module(sd, cost_volume_masked, keyframe, feats, prefix="depth_module"):
    x = torch.cat([cost_volume_masked, keyframe], 1)                                # :531
    cv_feats = []
    for i, (_, s) in enumerate(_DEPTH_ENC):
        x = conv_relu2(sd, f"{prefix}.enc.{i}.0", x, s)
        x = conv_relu2(sd, f"{prefix}.enc.{i}.1", x, 1)
        cv_feats.append(x)

    def head(idx, t):                                                               # :554-557
        w, b2 = sd[f"{prefix}.predictors.{idx}.1.weight"], sd[f"{prefix}.predictors.{idx}.1.bias"]
        return torch.abs(torch.tanh(conv_same(t, w, b2)))

    preds = []
    x = refine(sd, f"{prefix}.dec.0", cv_feats[4])                                  # i=0
    preds.insert(0, head(0, x))
    x = torch.cat([cv_feats[3], feats[2], x], 1)                                    # i=1
    x = conv_relu2(sd, f"{prefix}.dec.1.1", refine(sd, f"{prefix}.dec.1.0", x))
    preds.insert(0, head(1, x))
    x = torch.cat([cv_feats[2], feats[1], x], 1)                                    # i=2
    x = conv_relu2(sd, f"{prefix}.dec.2.1", refine(sd, f"{prefix}.dec.2.0", x))
    preds.insert(0, head(2, x))
    x = torch.cat([cv_feats[1], feats[0], x], 1)                                    # i=3 (no prediction :547)
    x = refine(sd, f"{prefix}.dec.3", x)
    x = torch.cat([cv_feats[0], x], 1)                                              # i=4
    x = conv_relu2(sd, f"{prefix}.dec.4.0", x)
    x = lrelu(conv_same(x, sd[f"{prefix}.dec.4.2.weight"], sd[f"{prefix}.dec.4.2.bias"]))
    preds.insert(0, head(3, x))
    return preds


# ----------------------------------------------------------------------------------------
# MonoRecModel.forward (monorec_model.py:672-729)
# ----------------------------------------------------------------------------------------
def forward(sd, batch, inv_depth_min_max=(0.33, 0.0025), cv_depth_steps=32, stages=None, use_ssim=True, cv_depths=None,
            sfcv_mult_mask=True, pretrain_mode=0, no_cv=False, mask_use_cv=True, mask_use_feats=True, simple_mask=False,
            cv_patch_size=3):
    """Returns the reference's output dict entries for eval mode (monorec_model.py:672-729); `pretrain_mode` as in :693-727."""
    with torch.no_grad():
        kf = batch["keyframe"]
        if not no_cv:                                                               # :680-686
            cv, sfcvs = cost_volume(batch, inv_depth_min_max[0], inv_depth_min_max[1], cv_depth_steps, patch_size=cv_patch_size, stages=stages,
                                    use_ssim=use_ssim, cv_depths=cv_depths, sfcv_mult_mask=sfcv_mult_mask)
        else:
            cv = kf.new_zeros(kf.shape[0], cv_depth_steps, kf.shape[2], kf.shape[3])
            sfcvs = [cv.clone() for _ in batch["poses"]]
        feats = resnet_features(sd, kf + .5)                                        # :691
        if pretrain_mode in (0, 2):
            if simple_mask:                                                         # :625-626; reads a previous prediction, :453
                cv_mask = mask_module(sd, sfcvs, feats, simple_input=(kf, batch["predicted_inverse_depths"][0]))
            else:
                cv_mask = mask_module(sd, sfcvs, feats, use_cv=mask_use_cv, use_features=mask_use_feats)   # :694
        elif pretrain_mode == 1:
            cv_mask = kf.new_zeros(kf.shape[0], 1, kf.shape[2], kf.shape[3])        # :708 (eval branch)
        else:
            cv_mask = batch["mvobj_mask"].clone()                                   # :711
        out = {"cost_volume_unmasked": cv, "single_frame_cvs": sfcvs, "image_features": feats, "cv_mask": cv_mask}
        if pretrain_mode == 2:                                                      # :723-724
            out["cost_volume"] = cv
            out["result"] = cv_mask
            return out
        cv_masked = (1 - cv_mask) * cv                                              # :713
        preds = depth_module(sd, cv_masked, kf, feats)                              # :715
        lo, hi = inv_depth_min_max[1], inv_depth_min_max[0]
        preds = [(1 - p) * lo + p * hi for p in preds]                              # :717-718
    out.update({"cost_volume": cv_masked, "predicted_inverse_depths": preds, "result": preds[0], "mask": cv_mask})
    return out


# ----------------------------------------------------------------------------------------
# sparse depth metrics (model/metric_functions/sparse_metrics.py:136-252, utils/util.py:36-118)
# ----------------------------------------------------------------------------------------
def _masked_mean(t, m, dim=None):
    """utils.mask_mean (utils/util.py:110-118)."""
    t = t.clone()
    t[m] = 0
    dims = list(range(t.dim())) if dim is None else dim
    els = 1
    for d in dims:
        els *= t.shape[d]
    return torch.sum(t, dim=dims) / (els - torch.sum(m.to(torch.float), dim=dims))


def sparse_metrics(pred, gt, roi=None, max_distance=None):
    """The seven metrics of configs/evaluate/eval_monorec.json:53-61 on inverse-depth maps (B,1,H,W),
    pred_all_valid=True, use_cvmask=False.  Returns a dict name -> 0-dim tensor."""
    if roi is not None:                                                    # preprocess_roi, util.py:36-43
        pred, gt = pred[:, :, roi[0]:roi[1], roi[2]:roi[3]], gt[:, :, roi[0]:roi[1], roi[2]:roi[3]]
    mask = gt == 0                                                         # get_mask, util.py:101-107
    if max_distance:
        mask = mask | (gt < 1 / max_distance)
    p, g = torch.relu(pred), torch.relu(gt)                                # get_positive_depth, util.py:59-65
    if max_distance is not None:                                           # get_absolute_depth, util.py:46-56
        p, g = torch.clamp_min(p, 1 / max_distance), torch.clamp_min(g, 1 / max_distance)
    dp, dg = 1 / p, 1 / g
    out = {}
    out["abs_rel_sparse_metric"] = _masked_mean(torch.abs(dp - dg) / dg, mask)            # :247-248
    out["sq_rel_sparse_metric"] = _masked_mean(((dp - dg) ** 2) / dg, mask)               # :251-252
    thresh = torch.max(dg / dp, dp / dg)
    out["a1_sparse_metric"] = _masked_mean((thresh < 1.25).float(), mask)                 # :205-207
    dp1, dg1 = dp.clone(), dg.clone()
    dp1[mask] = 1
    dg1[mask] = 1
    thresh1 = torch.max(dg1 / dp1, dp1 / dg1).float()
    out["a2_sparse_metric"] = _masked_mean((thresh1 < 1.25 ** 2).float(), mask)           # :210-214
    out["a3_sparse_metric"] = _masked_mean((thresh1 < 1.25 ** 3).float(), mask)           # :217-221
    out["rmse_sparse_metric"] = torch.mean(torch.sqrt(_masked_mean((dp1 - dg1) ** 2, mask, dim=[1, 2, 3])))   # :224-228
    out["rmse_log_sparse_metric"] = torch.mean(torch.sqrt(
        _masked_mean((torch.log(dp1) - torch.log(dg1)) ** 2, mask, dim=[1, 2, 3])))       # :231-235
    return out


# ----------------------------------------------------------------------------------------
# point-cloud path: create_pointcloud.py:76-94, utils/ply_utils.py:34-53, model/layers.py:43-61
# ----------------------------------------------------------------------------------------
def static_mask(cv_mask, mask_fill=32, threshold=0.1):
    """create_pointcloud.py:76-77: 1 where no pixel with cv_mask >= threshold lies in the (mask_fill+1)^2 window."""
    moving = (cv_mask >= threshold).to(torch.float32)
    box = F.conv2d(moving, moving.new_ones((1, 1, mask_fill + 1, mask_fill + 1)), padding=mask_fill // 2)
    return (box < 1).to(torch.float32)


def vote_mask(static_masks, min_hits=1):
    """create_pointcloud.py:90: a pixel survives when more than len - min_hits of the buffered masks are 1."""
    return (torch.sum(torch.stack(static_masks), dim=0) > len(static_masks) - min_hits).to(torch.float32)


def pointcloud_records(inv_depth, image, intrinsics, extrinsics, min_d=3, max_d=400, roi=None, dropout=0.0,
                       uniform=None, static_masks=None, min_hits=1):
    """(N, 6) x y z r g b records in the order PLYSaver.add_depthmap appends them (utils/ply_utils.py:34-53).
    `uniform` stands for the torch.rand_like(depth) of :45."""
    b, _, h, w = inv_depth.shape
    if static_masks:
        inv_depth = inv_depth * vote_mask(static_masks, min_hits)                          # create_pointcloud.py:91-92
    depth = 1 / inv_depth                                                                  # :36
    colour = (image + .5) * 255                                                            # :37
    keep = (min_d <= depth) & (depth <= max_d)                                             # :38
    if roi is not None:                                                                    # :39-43
        keep[:, :, :roi[0], :] = False
        keep[:, :, roi[1]:, :] = False
        keep[:, :, :, :roi[2]] = False
        keep[:, :, :, roi[3]:] = False
    if dropout > 0:                                                                        # :44-45
        keep = keep & (uniform > dropout)
    yy, xx = torch.meshgrid([torch.arange(0., float(h)), torch.arange(0., float(w))], indexing="ij")   # layers.py:49-54
    pix = torch.stack([xx.reshape(-1), yy.reshape(-1), torch.ones(h * w)], 0).unsqueeze(0).repeat(b, 1, 1)
    rays = torch.matmul(torch.inverse(intrinsics)[:, :3, :3], pix)                         # layers.py:57
    cam = depth.view(b, 1, -1) * rays                                                      # :58
    cam_h = torch.cat([cam, torch.ones(b, 1, h * w)], 1)                                   # :59
    world = (extrinsics @ cam_h)[:, :3, :]                                                 # ply_utils.py:48-49
    rec = torch.cat([world, colour.view_as(world)], dim=1).permute(0, 2, 1)                # :50
    return rec[keep.view(b, 1, -1).permute(0, 2, 1).expand(-1, -1, 6)].view(-1, 6)        # :51
