"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's `KittiOdometryDataset` sample assembly (data_loader/kitti_odometry_dataset.py:16-311)
for the sparse-target configurations (annotated lidar and / or D(V)SO depth), built on `oracle/input_oracle.py`.
The sequence metadata comes from the third-party `pykitti.odometry` in the reference (un-vendored pip dependency,
un-pinned); its published behaviour for the fields used here - calib.txt "P0".."P3" as 3x4 `P_rect_i0`, sorted
`image_i/*.png` lists, one 3x4 pose per line of `<pose_path>/<sequence>.txt` completed to 4x4, stereo baselines from
the x-shifts `P_i[0,3] / P_i[0,0]` - is restated in `read_sequence`.  Pinned in oracle/make_golden.py against the
unmodified reference class on a synthetic KITTI tree (monorec_amd.synth.make_kitti_tree) over the option matrix."""
import json
import os

import numpy as np
import torch
from PIL import Image

from . import input_oracle as io_


def read_sequence(root, seq, pose_folder):
    sdir = os.path.join(root, "sequences", seq)
    calib = {}
    for line in open(os.path.join(sdir, "calib.txt")):
        if ":" in line:
            name, vals = line.split(":", 1)
            calib[name.strip()] = np.array(vals.split(), dtype=np.float64)
    P = [calib[f"P{i}"].reshape(3, 4) for i in range(4)]
    files = []
    for cam in range(4):
        d = os.path.join(sdir, f"image_{cam}")
        files.append([os.path.join(d, n) for n in sorted(os.listdir(d))] if os.path.isdir(d) else [])
    poses = []
    ppath = os.path.join(root, pose_folder, seq + ".txt")
    if os.path.exists(ppath):
        for line in open(ppath):
            row = np.array(line.split(), dtype=np.float64)
            if row.size == 12:
                poses.append(np.concatenate([row.reshape(3, 4), np.array([[0, 0, 0, 1.0]])]))
    tx = [p[0, 3] / p[0, 0] for p in P]
    return dict(P=P, files=files, poses=poses, b_gray=abs(tx[1] - tx[0]), b_rgb=abs(tx[3] - tx[2]), dir=sdir)


def target_intrinsics(P_cam, orig, target):
    """compute_target_intrinsics (:318-349) -> fractional (f_x, f_y, c_x, c_y)."""
    r_orig, r_target = orig[0] / orig[1], target[0] / target[1]
    if r_orig >= r_target:
        new_h = r_target * orig[1]
        c_x, c_y = P_cam[0, 2] / orig[1], (P_cam[1, 2] - (orig[0] - new_h) / 2) / new_h
        rescale = orig[1] / target[1]
    else:
        new_w = orig[0] / r_target
        c_x, c_y = (P_cam[0, 2] - (orig[1] - new_w) / 2) / new_w, P_cam[1, 2] / orig[0]
        rescale = orig[0] / target[0]
    return P_cam[0, 0] / target[1] / rescale, P_cam[1, 1] / target[0] / rescale, c_x, c_y


def intrinsics_matrix(fr, target):
    m = torch.zeros(4, 4)                                                        # format_intrinsics, :366-375
    m[0, 0], m[1, 1], m[0, 2], m[1, 2], m[2, 2], m[3, 3] = fr[0] * target[1], fr[1] * target[0], fr[2] * target[1], fr[3] * target[0], 1, 1
    return m


class OracleKitti:
    def __init__(self, root, frame_count=2, sequences=None, depth_folder="image_depth", target_image_size=(256, 512),
                 max_length=None, dilation=1, offset_d=0, use_color=True, use_dso_poses=False, lidar_depth=False, dso_depth=True,
                 annotated_lidar=True, return_stereo=False, return_mvobj_mask=False, use_index_mask=()):
        assert (lidar_depth and annotated_lidar) or dso_depth, "sparse targets only"
        self.o = dict(root=str(root), fc=frame_count, seqs=list(sequences or [f"{i:02d}" for i in range(11)]), folder=depth_folder,
                      size=tuple(target_image_size), dil=dilation, off_d=offset_d, color=use_color, lidar=lidar_depth, dso=dso_depth,
                      stereo=return_stereo, mvobj=return_mvobj_mask, mask=use_index_mask)
        o = self.o
        self.seq = [read_sequence(o["root"], s, "poses_dvso" if use_dso_poses else "poses") for s in o["seqs"]]
        cam = 2 if use_color else 0
        self.cam = cam
        self.offset = (frame_count // 2) * dilation                               # :59
        extra = frame_count * dilation                                            # :60
        if annotated_lidar and lidar_depth:                                       # :61-63
            extra, self.offset = max(extra, 10), max(self.offset, 5)
        sizes = [len(s["files"][cam]) - (extra if use_index_mask is None else 0) for s in self.seq]    # :64-66
        self.indices = None
        if use_index_mask is not None:                                            # :67-84
            self.indices = []
            for n, name in zip(sizes, o["seqs"]):
                alive = {i: True for i in range(n)}
                for mask_name in use_index_mask:
                    m = json.load(open(os.path.join(o["root"], "sequences", name, mask_name + ".json")))
                    for k in list(alive):
                        if str(k) not in m or not m[str(k)]:
                            del alive[k]
                self.indices.append([k for k in sorted(alive) if self.offset <= k < n + self.offset - extra])
            sizes = [len(ix) for ix in self.indices]
        if max_length is not None:                                                # :85-86
            sizes = [min(n, max_length) for n in sizes]
        self.sizes = sizes
        self.orig = [tuple(reversed(Image.open(s["files"][cam][0]).size)) for s in self.seq]
        self.boxes = [io_.crop_box_for(h, w, *o["size"]) for h, w in self.orig]
        self.K = [intrinsics_matrix(target_intrinsics(s["P"][cam], og, o["size"]), o["size"]) for s, og in zip(self.seq, self.orig)]
        if dso_depth:                                                             # :351-355
            self.dso_par = [(*tuple(reversed(Image.open(s["files"][2][0]).size)), s["P"][2][0, 0]) for s in self.seq]

    def __len__(self):
        return sum(self.sizes)

    def image(self, di, cam, idx):
        return io_.preprocess_image(np.asarray(Image.open(self.seq[di]["files"][cam][idx])), self.boxes[di], *self.o["size"])

    def __getitem__(self, index):
        o = self.o
        di = 0
        while di < len(self.sizes) and index >= self.sizes[di]:                   # get_dataset_index, :111-117
            index -= self.sizes[di]
            di += 1
        if di == len(self.sizes):
            raise IndexError()
        if self.indices is not None:
            index = self.indices[di][index] - self.offset                         # :218-219
        s, key = self.seq[di], index + self.offset
        png = np.asarray(Image.open(os.path.join(s["dir"], o["folder"], f"{key:06d}.png"))).astype(np.uint16)
        if o["lidar"]:                                                            # :231-246
            depth = io_.lidar_inverse_depth(png, self.boxes[di], *o["size"]).unsqueeze(0)
        else:
            depth = torch.zeros(1, *o["size"])
        if o["dso"]:
            dso = io_.dso_inverse_depth(png, self.dso_par[di], self.boxes[di], *o["size"]).unsqueeze(0)
            hole = dso == 0
            dso[hole] = depth[hole]
            depth = dso
        steps = [i for i in range(-(o["fc"] // 2) * o["dil"], ((o["fc"] + 1) // 2) * o["dil"] + 1, o["dil"]) if i != 0]     # :254-255
        as_pose = lambda j: torch.tensor(s["poses"][j], dtype=torch.float32)
        data = {"keyframe": self.image(di, self.cam, key), "keyframe_pose": as_pose(key), "keyframe_intrinsics": self.K[di],
                "frames": [self.image(di, self.cam, key + i + o["off_d"]) for i in steps],
                "poses": [as_pose(key + i + o["off_d"]) for i in steps],
                "intrinsics": [self.K[di] for _ in range(o["fc"])],
                "sequence": torch.tensor([int(o["seqs"][di])], dtype=torch.int32),
                "image_id": torch.tensor([int(key)], dtype=torch.int32)}
        if o["stereo"]:                                                           # :272-279
            st = torch.eye(4)
            st[0, 3] = s["b_rgb"] if o["color"] else s["b_gray"]
            data["stereoframe"] = self.image(di, self.cam + 1, key)
            data["stereoframe_pose"] = as_pose(key) @ st
            data["stereoframe_intrinsics"] = self.K[di]
        if o["mvobj"] > 0:                                                        # :281-285
            mask = torch.tensor(np.load(os.path.join(s["dir"], "mvobj_mask", f"{key:06d}.npy")), dtype=torch.float32).unsqueeze(0)
            data["mvobj_mask"] = mask
            if o["mvobj"] == 2:
                return data, mask
        return data, depth
