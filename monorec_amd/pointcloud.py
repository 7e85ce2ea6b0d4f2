"""Point-cloud path on the MI355X: drop-in for the reference's `utils.PLYSaver`
(utils/ply_utils.py:8-53) and the mask logic of `create_pointcloud.py:57-102`.

    from monorec_amd.pointcloud import PLYSaver, static_mask, PointcloudBuilder

`PLYSaver` keeps the reference's constructor, `add_depthmap(depth, image, intrinsics, extrinsics)`, `save(file)`,
`.to(device)` and `.data` (an `array('f')` of x y z r g b records).  Differences that matter on the device:
  * the records stay in HBM behind a device-side cursor: `add_depthmap` is one asynchronous launch, the host copy
    happens once, in `save()` / on first access of `.data` (the reference does `.cpu().tolist()` per keyframe);
  * the 1/depth, range / roi / dropout tests, back-projection, pose multiply, colours and the ordered boolean-mask
    compaction are one kernel (`mr_pointcloud_append_f32`), optionally with the 5-mask vote and `depth *= mask`
    of create_pointcloud.py:90-92 fused in (`static_masks=`).
There is no CPU fallback: CPU tensors raise."""
import ctypes
from array import array

import torch

from . import _lib


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"monorec_amd.pointcloud: {what} must be a CUDA (HIP) tensor - there is no CPU fallback")


def static_mask(cv_mask, mask_fill=32, threshold=0.1):
    """create_pointcloud.py:76-77: `(F.conv2d((cv_mask >= .1).float(), ones(33, 33), padding=16) < 1).float()`."""
    _need_cuda(cv_mask, "cv_mask")
    lib = _lib.load()
    x = cv_mask.contiguous().float()
    b, c, h, w = x.shape
    assert c == 1
    out = torch.empty_like(x)
    _lib.check(lib.mr_static_mask_f32(x.data_ptr(), out.data_ptr(), b, h, w, float(threshold), int(mask_fill),
                                      torch.cuda.current_stream().cuda_stream), "mr_static_mask_f32")
    return out


class PLYSaver(torch.nn.Module):
    """utils/ply_utils.py:8-53 with device-resident records."""

    def __init__(self, height, width, min_d=3, max_d=400, batch_size=1, roi=None, dropout=0, capacity=1 << 22):
        super().__init__()
        self.height, self.width = height, width
        self.min_d, self.max_d = min_d, max_d
        self.roi = roi
        self.dropout = dropout
        self.batch_size = batch_size
        self._capacity = int(capacity)              # records; grows by doubling
        self._records = None                        # (capacity, 6) fp32 on the device
        self._cursor = None                         # int64[1] on the device
        self._host = array('f')                     # records already copied out (after save / .data)
        self._device = None
        self._pending = 0                           # upper bound of records appended since the last host sync

    # -- torch.nn.Module plumbing the reference script uses: plysaver.to(device)
    def _apply(self, fn):
        super()._apply(fn)
        probe = fn(torch.empty(0))
        self._device = probe.device
        return self

    def _ensure(self, device, incoming):
        if self._records is None:
            self._device = device
            self._records = torch.empty(self._capacity, 6, dtype=torch.float32, device=device)
            self._cursor = torch.zeros(1, dtype=torch.int64, device=device)
        if self._pending + incoming > self._capacity:           # could overflow: find out how full it really is
            used = int(self._cursor.item())
            self._pending = used
            while used + incoming > self._capacity:
                self._capacity *= 2
            if self._capacity > self._records.shape[0]:
                grown = torch.empty(self._capacity, 6, dtype=torch.float32, device=device)
                grown[:used] = self._records[:used]
                self._records = grown

    def add_depthmap(self, depth, image, intrinsics, extrinsics, static_masks=None, min_hits=1, uniform=None):
        """`depth` is the predicted inverse depth (B,1,H,W) like in the reference.  Extras: `static_masks` (list of the
        buffered static_mask() outputs) fuses the vote + `depth *= mask` of create_pointcloud.py:90-92; `uniform`
        replaces the `torch.rand_like(depth)` of ply_utils.py:45 (given: deterministic; None with dropout > 0: drawn
        here on the device)."""
        for t, n in ((depth, "depth"), (image, "image"), (intrinsics, "intrinsics"), (extrinsics, "extrinsics")):
            _need_cuda(t, n)
        lib = _lib.load()
        b, _, h, w = depth.shape
        self._ensure(depth.device, b * h * w)
        depth = depth.contiguous().float()
        image = image.contiguous().float()
        kinv = torch.inverse(intrinsics.float().cpu())[:, :3, :3].contiguous().to(depth.device)     # ply_utils.py:47, host LAPACK
        pose = extrinsics.float().contiguous()
        if self.dropout > 0 and uniform is None:
            uniform = torch.rand_like(depth)
        masks = list(static_masks) if static_masks else []
        ptrs = (ctypes.c_void_p * max(len(masks), 1))(*[m.contiguous().data_ptr() for m in masks])
        roi = (ctypes.c_int32 * 4)(*self.roi) if self.roi is not None else None
        _lib.check(lib.mr_pointcloud_append_f32(
            depth.data_ptr(), ptrs, len(masks), float(len(masks) - min_hits), image.data_ptr(), kinv.data_ptr(),
            pose.data_ptr(), uniform.contiguous().data_ptr() if uniform is not None else None, float(self.dropout),
            float(self.min_d), float(self.max_d), roi, b, h, w, self._records.data_ptr(), self._records.shape[0],
            self._cursor.data_ptr(), torch.cuda.current_stream().cuda_stream), "mr_pointcloud_append_f32")
        self._keep = (depth, image, kinv, pose, uniform, masks)          # alive until the launch has run
        self._pending += b * h * w

    def _drain(self):
        if self._records is None:
            return
        used = int(self._cursor.item())
        if used > self._records.shape[0]:
            raise RuntimeError("PLYSaver: device record buffer overflowed (internal capacity accounting)")
        if used:
            self._host.frombytes(self._records[:used].cpu().numpy().tobytes())
        self._cursor.zero_()
        self._pending = 0

    @property
    def data(self):
        self._drain()
        return self._host

    def save(self, file):
        """utils/ply_utils.py:18-32 (same header, binary little-endian float records)."""
        data = self.data
        fields = ("x", "y", "z", "red", "green", "blue")
        lines = ["ply", "format binary_little_endian 1.0", f"element vertex {len(data) // 6}"]
        lines += [f"property float {name}" for name in fields] + ["end_header"]
        file.write(("\n".join(lines) + "\n").encode("ascii"))
        data.tofile(file)


class PointcloudBuilder:
    """The keyframe loop body of create_pointcloud.py:66-102: static mask per keyframe, buffers of `buffer_length`,
    vote + masked depth of the middle keyframe into the PLYSaver.  Feed it the model's input/output dicts."""

    def __init__(self, plysaver, mask_fill=32, buffer_length=5, min_hits=1, use_mask=True):
        self.ply = plysaver
        self.mask_fill, self.buffer_length, self.min_hits, self.use_mask = mask_fill, buffer_length, min_hits, use_mask
        self.key_index = buffer_length // 2
        self._buf = []

    def add(self, data, result):
        output = result["result"]
        cv_mask = result["cv_mask"] if "cv_mask" in result else output.new_zeros(output.shape)
        # clones: `result` may come from submit() (views of resident buffers that later forwards overwrite), and the buffered
        # depth is multiplied in place by the vote below
        self._buf.append(dict(pose=data["keyframe_pose"].clone(), intrinsics=data["keyframe_intrinsics"].clone(),
                              mask=static_mask(cv_mask, self.mask_fill), keyframe=data["keyframe"].clone(),
                              depth=output.clone()))
        if len(self._buf) >= self.buffer_length:
            k = self._buf[self.key_index]
            masks = [e["mask"] for e in self._buf] if self.use_mask else None
            self.ply.add_depthmap(k["depth"], k["keyframe"], k["intrinsics"], k["pose"], static_masks=masks,
                                  min_hits=self.min_hits)
            del self._buf[0]
