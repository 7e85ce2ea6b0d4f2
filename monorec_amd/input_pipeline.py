"""Input pipeline on the MI355X: the per-frame work of the reference's
`KittiOdometryDataset` (data_loader/kitti_odometry_dataset.py) between the decoded image and the model input.

    crop box / intrinsics of the target size   compute_target_intrinsics, format_intrinsics  (:318-349, :366-375)
    crop -> PIL bilinear resize -> /255 - .5 -> CHW    preprocess_image (:120-134)      -> `ImagePreprocessor` (one launch)
    keyframe + neighbouring frames per sample  __getitem__ (:248-258)                  -> `FrameCache.sample`

The reference decodes and resizes every image three times with `frame_count = 2` (once as keyframe, twice as a
source frame of its neighbours); `FrameCache` keeps the preprocessed frames of the last few indices in HBM, so a
sequential sweep decodes each PNG once and runs one resize launch per *new* frame.  PNG decoding stays on the host
(PIL, as in the reference; `FrameCache(workers=N)` decodes ahead on N threads); everything after it runs on the device, bit-identical to Pillow's integer resampling.
There is no CPU fallback for the resize."""
import ctypes
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib


def compute_target_intrinsics(p_cam, orig_size, target_image_size):
    """Centre crop to the target aspect ratio and the intrinsics of the cropped, resized image - the rule of
    kitti_odometry_dataset.py:318-349 (same floating-point expressions, so the numbers are bit-identical).
    p_cam: 3x4 rectified projection (numpy), orig_size / target_image_size: (H, W).
    Returns ((f_x, f_y, c_x, c_y) as fractions of the target size, crop box (x0, y0, x1, y1))."""
    src_h, src_w = orig_size
    dst_h, dst_w = target_image_size
    aspect = dst_h / dst_w
    if src_h / src_w >= aspect:                      # source too tall: keep all columns, drop rows top and bottom
        kept = aspect * src_w
        margin = (src_h - kept) // 2
        box = (0, margin, src_w, src_h - margin)
        centre = (p_cam[0, 2] / src_w, (p_cam[1, 2] - (src_h - kept) / 2) / kept)
        shrink = src_w / dst_w
    else:                                            # source too wide: keep all rows, drop columns left and right
        kept = src_h / aspect
        margin = (src_w - kept) // 2
        box = (margin, 0, src_w - margin, src_h)
        centre = ((p_cam[0, 2] - (src_w - kept) / 2) / kept, p_cam[1, 2] / src_h)
        shrink = src_h / dst_h
    focal = (p_cam[0, 0] / dst_w / shrink, p_cam[1, 1] / dst_h / shrink)
    return (focal[0], focal[1], centre[0], centre[1]), box


def format_intrinsics(intrinsics, target_image_size):
    """Fractional (f_x, f_y, c_x, c_y) -> 4x4 pixel intrinsics of the target size (kitti_odometry_dataset.py:366-375)."""
    f_x, f_y, c_x, c_y = intrinsics
    h, w = target_image_size
    return torch.tensor([[f_x * w, 0.0, c_x * w, 0.0],
                         [0.0, f_y * h, c_y * h, 0.0],
                         [0.0, 0.0, 1.0, 0.0],
                         [0.0, 0.0, 0.0, 1.0]], dtype=torch.float32)


def _axis_tables(lib, in_size, out_size):
    ks = int(lib.mr_resample_ksize_bilinear(0, in_size, out_size))
    if ks < 0:
        _lib.check(ks, "mr_resample_ksize_bilinear")
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ks), dtype=np.int32)
    _lib.check(lib.mr_resample_coeffs_bilinear(in_size, 0, in_size, out_size, bounds.ctypes.data, coeffs.ctypes.data),
               "mr_resample_coeffs_bilinear")
    return ks, bounds, coeffs


class ImagePreprocessor:
    """`preprocess_image` (kitti_odometry_dataset.py:120-134) for one source image size: crop box (as `Image.crop`
    rounds it), Pillow-exact bilinear resize to `target_image_size`, `/255 - .5`, CHW - one launch per image."""

    def __init__(self, orig_size, target_image_size, crop_box=None, device="cuda:0"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("monorec_amd.input_pipeline: a HIP device is required - there is no CPU fallback")
        self.orig_h, self.orig_w = int(orig_size[0]), int(orig_size[1])
        self.out_h, self.out_w = int(target_image_size[0]), int(target_image_size[1])
        if crop_box is None:
            crop_box = (0, 0, self.orig_w, self.orig_h)
        self.box = tuple(int(round(v)) for v in crop_box)                   # PIL: Image.crop rounds the box
        cw, ch = self.box[2] - self.box[0], self.box[3] - self.box[1]
        self.hks, hb, hk = _axis_tables(self.lib, cw, self.out_w)
        self.vks, vb, vk = _axis_tables(self.lib, ch, self.out_h)
        self.max_rows = max(int(vb[min(t + 15, self.out_h - 1), 0] + vb[min(t + 15, self.out_h - 1), 1] - vb[t, 0])
                            for t in range(0, self.out_h, 16))
        up = lambda a: torch.from_numpy(a).to(self.device)
        self.hb, self.hk, self.vb, self.vk = up(hb), up(hk), up(vb), up(vk)
        self._box_c = (ctypes.c_int32 * 4)(*self.box)
        self._staging = {}          # channels -> ring of (pinned host buffer, device buffer, copy-done event)
        self._ring_pos = 0

    def __call__(self, image, out=None):
        """image: uint8 (H, W, 3) or (H, W), numpy / CPU tensor (uploaded) or device tensor -> float32 (3, h, w) on the device."""
        if torch.is_tensor(image) and not image.is_cuda:
            image = image.numpy()
        if isinstance(image, np.ndarray):
            if image.dtype != np.uint8 or image.ndim not in (2, 3):
                raise ValueError("expected a uint8 (H, W[, 3]) image")
            channels = 1 if image.ndim == 2 else image.shape[2]
        else:
            if image.dtype != torch.uint8 or image.dim() not in (2, 3):
                raise ValueError("expected a uint8 (H, W[, 3]) image")
            channels = 1 if image.dim() == 2 else image.shape[2]
        if (image.shape[0], image.shape[1]) != (self.orig_h, self.orig_w) or channels not in (1, 3):
            raise ValueError(f"image of shape {tuple(image.shape)} does not match this preprocessor ({self.orig_h}x{self.orig_w})")
        if isinstance(image, np.ndarray):
            # host image: through a small ring of pinned staging buffers (allocated once), asynchronous upload
            ring = self._staging.get(channels)
            if ring is None:
                shape = (self.orig_h, self.orig_w) if channels == 1 else (self.orig_h, self.orig_w, channels)
                ring = [(torch.empty(shape, dtype=torch.uint8).pin_memory(), torch.empty(shape, dtype=torch.uint8, device=self.device),
                         torch.cuda.Event()) for _ in range(4)]
                self._staging[channels] = ring
            host, devbuf, ev = ring[self._ring_pos % len(ring)]
            self._ring_pos += 1
            ev.synchronize()                                   # the upload that last used this slot has finished
            host.numpy()[...] = image
            devbuf.copy_(host, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
            image = devbuf
        image = image.contiguous()
        if out is None:
            out = torch.empty(3, self.out_h, self.out_w, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mr_preprocess_image_u8_f32(
            image.data_ptr(), self.orig_h, self.orig_w, channels, self.orig_w * channels, self._box_c, self.out_h, self.out_w,
            self.hb.data_ptr(), self.hk.data_ptr(), self.hks, self.vb.data_ptr(), self.vk.data_ptr(), self.vks,
            self.max_rows, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "mr_preprocess_image_u8_f32")
        return out


def lidar_inverse_depth(depth_png, crop_box, target_image_size, device="cuda:0"):
    """`preprocess_depth_annotated_lidar` (kitti_odometry_dataset.py:184-211): uint16 depth PNG array (H, W), numpy / tensor
    -> sparse inverse-depth target (target_h, target_w) float32 on the device (add the channel dim like `:238`)."""
    lib = _lib.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("monorec_amd.input_pipeline: a HIP device is required - there is no CPU fallback")
    if isinstance(depth_png, np.ndarray):
        if depth_png.dtype != np.uint16:
            raise ValueError("expected the 16-bit depth PNG as a uint16 array")
        depth_png = torch.from_numpy(np.ascontiguousarray(depth_png).view(np.int16))
    if depth_png.dtype not in (torch.int16, torch.uint16) or depth_png.dim() != 2:
        raise ValueError("expected a (H, W) 16-bit image")
    src = depth_png.contiguous().to(device, non_blocking=True)
    h, w = src.shape
    th, tw = int(target_image_size[0]), int(target_image_size[1])
    box = None if crop_box is None else (ctypes.c_int32 * 4)(*[int(v) for v in crop_box])
    owner = torch.empty(th * tw, dtype=torch.int32, device=device)
    out = torch.empty(th, tw, dtype=torch.float32, device=device)
    _lib.check(lib.mr_lidar_inverse_depth_u16_f32(src.data_ptr(), h, w, box, th, tw, owner.data_ptr(), out.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "mr_lidar_inverse_depth_u16_f32")
    return out


def dso_inverse_depth(depth_png, dso_depth_parameters, crop_box, target_image_size, device="cuda:0"):
    """`preprocess_depth_dso` (kitti_odometry_dataset.py:156-182): uint16 D(V)SO depth PNG array -> sparse inverse-depth map
    (target_h, target_w) float32 on the device.  `dso_depth_parameters` = (original image height, width, f_x) (:351-355)."""
    lib = _lib.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("monorec_amd.input_pipeline: a HIP device is required - there is no CPU fallback")
    if isinstance(depth_png, np.ndarray):
        if depth_png.dtype != np.uint16:
            raise ValueError("expected the 16-bit depth PNG as a uint16 array")
        depth_png = torch.from_numpy(np.ascontiguousarray(depth_png).view(np.int16))
    if depth_png.dtype not in (torch.int16, torch.uint16) or depth_png.dim() != 2:
        raise ValueError("expected a (H, W) 16-bit image")
    src = depth_png.contiguous().to(device, non_blocking=True)
    h, w = src.shape
    oh, ow, f_x = dso_depth_parameters
    th, tw = int(target_image_size[0]), int(target_image_size[1])
    box = None if crop_box is None else (ctypes.c_int32 * 4)(*[int(v) for v in crop_box])
    owner = torch.empty(th * tw, dtype=torch.int32, device=device)
    out = torch.empty(th, tw, dtype=torch.float32, device=device)
    _lib.check(lib.mr_dso_inverse_depth_u16_f32(src.data_ptr(), h, w, int(oh), int(ow), float(f_x), box, th, tw, owner.data_ptr(),
                                                out.data_ptr(), torch.cuda.current_stream().cuda_stream), "mr_dso_inverse_depth_u16_f32")
    return out


class FrameCache:
    """Preprocessed frames by index, least recently used evicted.  `load(index)` returns the decoded uint8 image
    (the reference: `dataset.get_cam2(index)` / `get_cam0`, a PIL image - `np.asarray` of it works).

    `workers > 0` decodes ahead on that many host threads (PIL's PNG decoder releases the GIL), the counterpart of the
    reference's `num_workers` data-loader processes (configs/evaluate/eval_monorec.json:33): whenever frame i is asked
    for, the decodes of i+1 .. i+lookahead are started, so a sequential sweep finds its next image already decoded.
    Only `load` runs on the threads; the device launches stay on the caller's thread and stream.  `index_range = (lo, hi)`
    bounds the read-ahead (hi exclusive)."""

    def __init__(self, load, preprocessor, capacity=8, workers=0, lookahead=None, index_range=None):
        self.load, self.pre, self.capacity = load, preprocessor, capacity
        self._frames = OrderedDict()
        self.decoded = 0                       # number of images decoded + resized so far
        self.workers = int(workers)
        self.lookahead = int(lookahead if lookahead is not None else 2 * self.workers)
        self.index_range = index_range
        self._pool = ThreadPoolExecutor(self.workers, thread_name_prefix="monorec-decode") if self.workers > 0 else None
        self._decoding = {}                    # index -> Future of the decoded image

    def _decode_host(self, index):
        return np.ascontiguousarray(np.asarray(self.load(index)))

    def prefetch(self, indices):
        """Start decoding `indices` on the worker threads (no-op without workers, for cached or already started ones)."""
        if self._pool is None:
            return
        for j in indices:
            if j in self._frames or j in self._decoding:
                continue
            if self.index_range is not None and not (self.index_range[0] <= j < self.index_range[1]):
                continue
            self._decoding[j] = self._pool.submit(self._decode_host, j)

    def frame(self, index):
        if index in self._frames:
            self._frames.move_to_end(index)
            return self._frames[index]
        fut = self._decoding.pop(index, None)
        img = fut.result() if fut is not None else self._decode_host(index)
        t = self.pre(img)
        self.decoded += 1
        self._frames[index] = t
        while len(self._frames) > self.capacity:
            self._frames.popitem(last=False)
        if self._pool is not None:
            self.prefetch(range(index + 1, index + 1 + self.lookahead))
            for j in [j for j in self._decoding if j < index - self.capacity]:     # read-ahead the sweep has left behind
                self._decoding.pop(j).cancel()
        return t

    def sample(self, index, frame_count=2, dilation=1, offset_d=0):
        """keyframe and source frames of `__getitem__` (kitti_odometry_dataset.py:248-255): returns
        (keyframe (3,H,W), [frames], [source indices])."""
        offs = [i for i in range(-(frame_count // 2) * dilation, ((frame_count + 1) // 2) * dilation + 1, dilation) if i != 0]
        idx = [index + i + offset_d for i in offs]
        return self.frame(index), [self.frame(j) for j in idx], idx

    def close(self):
        if self._pool is not None:
            for fut in self._decoding.values():
                fut.cancel()
            self._decoding.clear()
            self._pool.shutdown(wait=True)
            self._pool = None
