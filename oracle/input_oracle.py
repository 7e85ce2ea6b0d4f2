"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's image preprocessing:
`KittiOdometryDataset.preprocess_image` (data_loader/kitti_odometry_dataset.py:120-134) =
    img.crop(box) -> img.resize((W, H), resample=Image.BILINEAR) -> float32 / 255 - .5 -> CHW
and the crop box of `compute_target_intrinsics` (:318-349).

The resampling itself lives in a third-party dependency that is absent from /root/reference: **Pillow**
(un-pinned in environment.yml; pulled in by torchvision, python 3.6 era => Pillow <= 8.4).  Its published algorithm
(`src/libImaging/Resample.c`, unchanged in arithmetic since Pillow 3.4) is restated here:
  * per output coordinate xx: center = in0 + (xx + .5) * scale, support = max(scale, 1) (bilinear: filter support 1),
    xmin = max(0, int(center - support + .5)), xmax = min(in_size, int(center + support + .5)),
    triangle weights w(x) = max(0, 1 - |(x + xmin - center + .5) / max(scale, 1)|) normalised to sum 1 (double);
  * 8-bit images: weights -> fixed point, k = int(.5 + w * 2^22); pixel = clip8((2^21 + sum pixel * k) >> 22);
  * two passes, horizontal first (into an 8-bit intermediate of only the rows the vertical pass needs), then vertical.
Pinned against the Pillow installed in this image (12.2) on seeded images and on the reference's example frames
(`oracle/make_golden.py`); parity is **bit-exact** (integer work).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def crop_box_for(orig_h, orig_w, target_h, target_w):
    """compute_target_intrinsics (kitti_odometry_dataset.py:322-343): centre crop to the target aspect ratio."""
    r_orig, r_target = orig_h / orig_w, target_h / target_w
    if r_orig >= r_target:
        new_height = r_target * orig_w
        return (0, (orig_h - new_height) // 2, orig_w, orig_h - (orig_h - new_height) // 2)
    new_width = orig_h / r_target
    return ((orig_w - new_width) // 2, 0, orig_w - (orig_w - new_width) // 2, orig_h)


def resample_coeffs(in_size, in0, in1, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter.
    Returns (ksize, bounds int32 (out_size, 2) = (first, count), coeffs int32 (out_size, ksize))."""
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w)                                   # C accumulates in this order too
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(w):
            coeffs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return ksize, bounds, coeffs


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow `Image.resize((out_w, out_h), Image.BILINEAR)` of an HxW or HxWxC uint8 array."""
    a = img[:, :, None] if img.ndim == 2 else img
    h, w, c = a.shape
    _, hb, hk = resample_coeffs(w, 0, w, out_w)
    _, vb, vk = resample_coeffs(h, 0, h, out_h)
    first, last = int(vb[0, 0]), int(vb[-1, 0] + vb[-1, 1])
    src = a[first:last].astype(np.int64)
    if out_w != w:
        tmp = np.empty((last - first, out_w, c), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = hb[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], hk[xx, :n].astype(np.int64), axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
    else:
        tmp = a[first:last]
    if out_h != h:
        out = np.empty((out_h, out_w, c), dtype=np.uint8)
        t64 = tmp.astype(np.int64)
        for yy in range(out_h):
            y0, n = vb[yy]
            y0 -= first
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(vk[yy, :n].astype(np.int64), t64[y0:y0 + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
    else:
        out = tmp
    return out[:, :, 0] if img.ndim == 2 else out


def preprocess_image(img_u8, crop_box, target_h, target_w):
    """kitti_odometry_dataset.py:120-134 on a decoded uint8 image (H,W,3 colour or H,W grey) -> float32 (3,H,W)."""
    import torch
    if crop_box is not None:
        x0, y0, x1, y1 = (int(round(v)) for v in crop_box)               # Image.crop rounds the box
        img_u8 = img_u8[y0:y1, x0:x1]
    out = resize_bilinear_u8(np.ascontiguousarray(img_u8), target_h, target_w)
    t = torch.tensor(out.astype(np.float32)) / 255 - .5                  # :127-128
    return torch.stack((t, t, t)) if t.dim() == 2 else t.permute(2, 0, 1)  # :129-132


def lidar_inverse_depth(depth_png, crop_box, target_h, target_w):
    """preprocess_depth_annotated_lidar (kitti_odometry_dataset.py:184-211) on the decoded uint16 PNG array."""
    import torch
    d = depth_png.astype(np.float64)
    h, w = d.shape
    rows, cols = np.nonzero(d)                                             # row-major order (:187)
    inv = 256.0 / d[rows, cols]                                            # :189-190
    pts = np.stack([rows.astype(np.float64), cols.astype(np.float64), inv])
    if crop_box is not None:                                               # :194-200
        x0, y0, x1, y1 = crop_box
        sel = (y0 <= pts[0]) & (pts[0] < y1) & (x0 <= pts[1]) & (pts[1] < x1)
        pts = pts[:, sel]
        pts[0] -= y0
        pts[1] -= x0
        ch, cw = y1 - y0, x1 - x0
    else:
        ch, cw = h, w
    pts[0] = np.clip(pts[0] / ch * target_h, 0, target_h - 1)              # :205
    pts[1] = np.clip(pts[1] / cw * target_w, 0, target_w - 1)              # :206
    out = np.zeros((target_h, target_w))
    out[np.around(pts[0]).astype(int), np.around(pts[1]).astype(int)] = pts[2]   # :209 (last write wins)
    return torch.tensor(out, dtype=torch.float32)


def dso_inverse_depth(depth_png, dso_depth_parameters, crop_box, target_h, target_w):
    """preprocess_depth_dso (kitti_odometry_dataset.py:156-182) on the decoded uint16 PNG array; `dso_depth_parameters`
    = (original image height, width, f_x) as get_dso_depth_parameters returns them (:351-355)."""
    import torch
    h, w, f_x = dso_depth_parameters
    d = depth_png.astype(np.float64)
    rows, cols = np.nonzero(d)
    pts = np.stack([np.clip(rows.astype(np.float64) / d.shape[0] * h, 0, h - 1),          # :160
                    np.clip(cols.astype(np.float64) / d.shape[1] * w, 0, w - 1),          # :161
                    w * d[rows, cols] / (0.54 * f_x * 65535)])                            # :163-164
    if crop_box is not None:                                                              # :168-174
        x0, y0, x1, y1 = crop_box
        pts = pts[:, (y0 <= pts[0]) & (pts[0] < y1) & (x0 <= pts[1]) & (pts[1] < x1)]
        pts[0] -= y0
        pts[1] -= x0
        ch, cw = y1 - y0, x1 - x0
    else:
        ch, cw = h, w
    pts[0] = np.clip(pts[0] / ch * target_h, 0, target_h - 1)                             # :178
    pts[1] = np.clip(pts[1] / cw * target_w, 0, target_w - 1)                             # :179
    out = np.zeros((target_h, target_w))
    out[np.around(pts[0]).astype(int), np.around(pts[1]).astype(int)] = pts[2]            # :182 (last write wins)
    return torch.tensor(out, dtype=torch.float32)

