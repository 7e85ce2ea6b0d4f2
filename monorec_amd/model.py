"""Drop-in host-side mirror of the reference model API for the cost-volume inference path.

`MonoRecModel` keeps the reference's constructor signature, public attributes, state-dict keys and
dict-in / dict-out `forward` (reference model/monorec/monorec_model.py:560-729), so
`evaluate.py`, `create_pointcloud.py` and `example/test_monorec.py` can use it unchanged
(INTEGRATION.md).  The sub-modules only *hold parameters* under the reference's names
(SURVEY.md Appendix C); all arithmetic runs in the gfx950 kernels of libmonorec_hip.so through the
launch plan in `engine.py`.  There is no PyTorch/CPU fallback: calling `forward` without a HIP device
or without the built library raises.
"""
import collections
import ctypes
import threading
import time
import warnings
import weakref

import numpy as np
import torch
from torch import nn

from . import _lib
from .engine import Plan

_METRIC_CACHE_KEY = "_monorec_amd_metric_sums"      # monorec_amd.metrics caches its fused sums on the dict under this key


class _ParamsOnly(nn.Module):
    """Container whose children exist to own parameters with reference-compatible names."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("monorec_amd sub-modules hold parameters only; call MonoRecModel.forward")


def _conv(cin, cout, k, stride=1, bias=True):
    return nn.Conv2d(cin, cout, k, stride=stride, bias=bias)


class _Named(_ParamsOnly):
    def __init__(self, **children):
        super().__init__()
        for n, m in children.items():
            self.add_module(n, m)


def _seq(*mods):
    s = nn.Sequential()
    for i, m in enumerate(mods):
        s.add_module(str(i), m)
    return s


class _Placeholder(_ParamsOnly):
    """Occupies a Sequential index that holds a parameter-free op in the reference (pool/pad/act)."""


# --------------------------------------------------------------------------------------------------
class ResnetEncoder(_ParamsOnly):
    """Parameters of ResnetEncoder (monorec_model.py:95-129): torchvision ResNet-18 key layout."""

    def __init__(self, num_layers=18, pretrained=True):
        super().__init__()
        if num_layers != 18:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        enc = _ParamsOnly()
        enc.conv1 = _conv(3, 64, 7, 2, bias=False)
        enc.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, cout in enumerate((64, 128, 256, 512), start=1):
            blocks = []
            for bi in range(2):
                stride = 2 if (li > 1 and bi == 0) else 1
                blk = _Named(conv1=_conv(cin, cout, 3, stride, bias=False), bn1=nn.BatchNorm2d(cout),
                             conv2=_conv(cout, cout, 3, 1, bias=False), bn2=nn.BatchNorm2d(cout))
                if stride != 1 or cin != cout:
                    blk.downsample = _seq(_conv(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
                blocks.append(blk)
                cin = cout
            setattr(enc, f"layer{li}", _seq(*blocks))
        enc.fc = nn.Linear(512, 1000)   # present in torchvision's state dict; unused by the path
        self.encoder = enc
        # monorec_model.py:104-113 builds torchvision.models.resnet18(pretrained): the ImageNet weights come from torchvision's
        # download cache.  Same here when torchvision is importable; otherwise the encoder stays randomly initialised and
        # MonoRecModel warns at its first forward unless a checkpoint / state dict supplied `_feature_extractor.*`.
        self._weights_loaded = False
        if pretrained:
            try:
                import torchvision
                tv = torchvision.models.resnet18(pretrained)
                enc.load_state_dict(tv.state_dict(), strict=True)
                self._weights_loaded = True
            except Exception:
                pass


class CostVolumeModule(_ParamsOnly):
    """Options of CostVolumeModule (monorec_model.py:132-148); no parameters."""

    def __init__(self, use_mono=True, use_stereo=False, use_ssim=True, patch_size=3,
                 channel_weights=(5 / 32, 16 / 32, 11 / 32), alpha=10, not_center_cv=False, sfcv_mult_mask=True):
        super().__init__()
        self.use_mono, self.use_stereo, self.use_ssim = use_mono, use_stereo, use_ssim
        self.patch_size = patch_size
        self.border_radius = patch_size // 2 + 1
        self.channel_weights = tuple(channel_weights)
        self.alpha = alpha
        self.not_center_cv = not_center_cv
        self.sfcv_mult_mask = sfcv_mult_mask


class MaskModule(_ParamsOnly):
    """Parameters of MaskModule (monorec_model.py:287-343)."""

    def __init__(self, depth_steps=32, feature_channels=(64, 64, 128, 256, 512), use_cv=True, use_features=True, in_channels=None):
        super().__init__()
        self.depth_steps = depth_steps
        self.feat_chns = feature_channels
        self.use_cv, self.use_features = use_cv, use_features
        ec = (depth_steps if in_channels is None else in_channels, 48, 64, 96, 96)
        dc = (96, 96, 64, 48)
        fc = tuple(int(c) for c in feature_channels)

        def cr(cin, cout):
            return _Named(conv=_conv(cin, cout, 3))

        enc = []
        for i in range(5):
            mods = [cr(ec[i - 1] if i else ec[0], ec[i]), cr(ec[i], ec[i])]
            enc.append(_seq(*mods) if i == 0 else _seq(nn.MaxPool2d(2), *mods))
        self.enc = nn.ModuleList(enc)
        up_in = (ec[4] + fc[3], dc[0], dc[1], dc[2])
        up_out = (dc[0], dc[0], dc[1], dc[2])
        cat_in = (up_out[0] + ec[3] + fc[2], up_out[1] + ec[2] + fc[1], up_out[2] + ec[1] + fc[0], up_out[3] + ec[0])
        dec = []
        for i in range(4):
            dec.append(_seq(_Named(conv=_conv(up_in[i], up_out[i], 2)), cr(cat_in[i], dc[i]), cr(dc[i], dc[i])))
        self.dec = nn.ModuleList(dec)
        self.classifier = _seq(_conv(dc[3], 1, 1), nn.Sigmoid())


class SimpleMaskModule(MaskModule):
    """Parameters of SimpleMaskModule (monorec_model.py:388-442): the MaskModule layout with depth_steps + 3 + 1 input channels."""

    def __init__(self, depth_steps=32, feature_channels=(64, 64, 128, 256, 512)):
        super().__init__(depth_steps, feature_channels, in_channels=depth_steps + 3 + 1)


class DepthModule(_ParamsOnly):
    """Parameters of DepthModule (monorec_model.py:476-524)."""

    def __init__(self, depth_steps=32, feature_channels=(64, 64, 128, 256, 512), large_model=False):
        super().__init__()
        self.depth_steps = depth_steps
        self.feat_chns = feature_channels
        fc = tuple(int(c) for c in feature_channels)
        ec = (48, 64, 128, 256, 512) if large_model else (48, 64, 128, 192, 256)          # monorec_model.py:482
        dc = (512, 256, 128, 64, 32, 24) if large_model else (256, 128, 64, 48, 32, 24)  # :483
        ks = (7, 7, 5, 5, 3)

        def cr2(cin, cout, k, s=1):
            return _Named(conv_y=nn.Conv2d(cin, cout, (k, 1), stride=(s, 1)),
                          conv_x=nn.Conv2d(cout, cout, (1, k), stride=(1, s)))

        def refine(cin, cout):
            return _Named(conv2d_t=nn.ConvTranspose2d(cin, cout, 4, stride=2))

        cin = depth_steps + 3
        enc = []
        for i in range(5):
            enc.append(_seq(cr2(cin, ec[i], ks[i], 1 if i == 0 else 2), cr2(ec[i], ec[i], 3)))
            cin = ec[i]
        self.enc = nn.ModuleList(enc)
        self.dec = nn.ModuleList([
            refine(ec[4], dc[0]),
            _seq(refine(ec[3] + fc[2] + dc[0], dc[1]), cr2(dc[1], dc[1], 3)),
            _seq(refine(ec[2] + fc[1] + dc[1], dc[2]), cr2(dc[2], dc[2], 3)),
            refine(ec[1] + fc[0] + dc[2], dc[3]),
            _seq(cr2(ec[0] + dc[3], dc[4], 3), _Placeholder(), _conv(dc[4], dc[5], 3), nn.LeakyReLU(0.1)),
        ])
        self.predictors = nn.ModuleList([_seq(_Placeholder(), _conv(c, 1, 3)) for c in dc[:3] + dc[-1:]])


# --------------------------------------------------------------------------------------------------
_KINV_CACHE = {}    # bytes of the (B, 4, 4) keyframe intrinsics -> kinv (B, 9): a dataset has one intrinsics matrix, every keyframe brings it again


def host_geometry(keyframe_intrinsics, keyframe_pose, intrinsics, poses):
    """The 4x4 algebra of CostVolumeModule.forward on CPU fp32 tensors, operation for operation
    (monorec_model.py:171 inverse(pose); :198 inverse(K_kf); :207 ext @ pose_kf; layers.py:65 K @ T).

    Done on the host with the same ATen CPU operators as the reference so that the matrices handed to
    the kernel are bit-identical to the reference's (the only ill-conditioned step of the path,
    SURVEY.md section 0).  Returns kinv (B,9) and proj (B,F,12).

    It sits on the critical path of every keyframe (the cost volume cannot start before it), so it is kept short: the F pose
    inversions are ONE batched call (the CPU kernel loops over the matrices with the same LAPACK routine - bit-identical to F
    calls, asserted in tests/test_capi_and_host.py) and the inverse of the keyframe intrinsics is remembered by content."""
    b = keyframe_pose.shape[0]
    nf = len(poses)
    key = keyframe_intrinsics.detach().contiguous().numpy().tobytes()
    kinv = _KINV_CACHE.get(key)
    if kinv is None:
        kinv = torch.empty(b, 9)
        for n in range(b):
            kinv[n] = torch.inverse(keyframe_intrinsics[n]).unsqueeze(0)[:, :3, :3].reshape(9)
        if len(_KINV_CACHE) >= 16:
            _KINV_CACHE.clear()
        _KINV_CACHE[key] = kinv
    proj = torch.empty(b, nf, 12)
    if nf:
        extr = torch.inverse(torch.stack([p for p in poses]))           # (F, B, 4, 4)
        for n in range(b):
            kp = keyframe_pose[n]
            for f in range(nf):
                t = extr[f, n] @ kp
                proj[n, f] = torch.matmul(intrinsics[f][n].unsqueeze(0), t.unsqueeze(0))[:, :3, :].reshape(12)
    return kinv, proj


def host_geometry_reference_form(keyframe_intrinsics, keyframe_pose, intrinsics, poses):
    """host_geometry as the reference writes it - one torch.inverse per matrix, no cache.  Test oracle for host_geometry only."""
    b = keyframe_pose.shape[0]
    nf = len(poses)
    extr = [torch.inverse(p) for p in poses]
    kinv = torch.empty(b, 9)
    proj = torch.empty(b, nf, 12)
    for n in range(b):
        kinv[n] = torch.inverse(keyframe_intrinsics[n]).unsqueeze(0)[:, :3, :3].reshape(9)
        for f in range(nf):
            t = extr[f][n] @ keyframe_pose[n]
            proj[n, f] = torch.matmul(intrinsics[f][n].unsqueeze(0), t.unsqueeze(0))[:, :3, :].reshape(12)
    return kinv, proj


HOST_SPIN_SECONDS = 0.004       # longest busy-poll of _host_wait before it falls back to a sleeping wait


import os as _os
_STREAM_LAYOUT = _os.environ.get("MR_DIAG_STREAM_LAYOUT")                  # diagnostic: creation / first-use order of the model's streams (MonoRecModel._device_streams)
_STREAM_PRIO = _os.environ.get("MR_DIAG_STREAM_PRIO", "")                  # diagnostic: kinds of streams ("m", "e", "g") created with high priority


def _host_wait(event):
    """Block the host until `event` has happened - by polling, for a bounded time.  hipEventSynchronize polls only for a while and
    then sleeps, and a sleeping wait wakes up late (measured on MI355X / ROCm 7: sequential forwards of 2 ms each went to 3.2 ms
    once the wait inside submit() crossed ~2 ms).  The waits of a keyframe stream are short (a c2 keyframe takes ~1.5 ms) and end
    with work for the host, so the host polls - the GIL is released between polls (nn.DataParallel drives replicas from threads) -
    but only for HOST_SPIN_SECONDS: behind that (batched workloads, a stalled device, eight ranks of a node each spinning a core
    for nothing) it hands the wait to hipEventSynchronize, whose late wake-up no longer matters at that length."""
    if event.query():
        return
    deadline = time.perf_counter() + HOST_SPIN_SECONDS
    while not event.query():
        if time.perf_counter() > deadline:
            event.synchronize()
            return
        time.sleep(0)


def depth_hypotheses(inv_depth_min_max, steps):
    """monorec_model.py:675-677,184: the bounds pass through fp32 tensors and `.item()` before linspace."""
    lo = torch.tensor([inv_depth_min_max[1]], dtype=torch.float32)[0].item()
    hi = torch.tensor([inv_depth_min_max[0]], dtype=torch.float32)[0].item()
    return 1 / torch.linspace(lo, hi, int(steps))


class MonoRecModel(nn.Module):
    """MI355X-native MonoRecModel: same constructor / attributes / state dict / forward contract as the
    reference class (monorec_model.py:560-729); inference (eval, pretrain_mode=0) only."""

    def __init__(self, inv_depth_min_max=(0.33, 0.0025), cv_depth_steps=32, pretrain_mode=False, pretrain_dropout=0.0,
                 pretrain_dropout_mode=0, augmentation=None, use_mono=True, use_stereo=False, use_ssim=True,
                 sfcv_mult_mask=True, simple_mask=False, mask_use_cv=True, mask_use_feats=True, cv_patch_size=3,
                 depth_large_model=False, no_cv=False, freeze_resnet=True, freeze_module=(), checkpoint_location=None,
                 mask_cp_loc=None, depth_cp_loc=None, hip_graph=False, hip_in_flight=4, hip_bf16=False, hip_bf16x3=False,
                 hip_batch_keyframes=1, hip_queue_depth=1, hip_single_stream=False, hip_exact_convs=False, hip_cv_separable=False, hip_lean_outputs=False, hip_skip_dead_layer4=False,
                 hip_slot_streams=None, hip_streams=None, hip_forward_on_callers_stream=True):
        super().__init__()
        self.inv_depth_min_max = inv_depth_min_max
        self.cv_depth_steps = cv_depth_steps
        self.use_mono = use_mono
        self.use_stereo = use_stereo
        self.use_ssim = use_ssim
        self.sfcv_mult_mask = sfcv_mult_mask
        self.pretrain_mode = int(pretrain_mode)
        self.pretrain_dropout = pretrain_dropout
        self.pretrain_dropout_mode = pretrain_dropout_mode
        self.augmentation = augmentation
        self.simple_mask = simple_mask
        self.mask_use_cv = mask_use_cv
        self.mask_use_feats = mask_use_feats
        self.cv_patch_size = cv_patch_size
        self.no_cv = no_cv
        self.depth_large_model = depth_large_model
        self.checkpoint_location = checkpoint_location
        self.mask_cp_loc = mask_cp_loc
        self.depth_cp_loc = depth_cp_loc
        self.freeze_module = freeze_module
        self.freeze_resnet = freeze_resnet
        unsupported = dict(pretrain_mode=self.pretrain_mode not in (0, 1, 2, 3), use_mono=not (use_mono or use_stereo or no_cv),
                           use_ssim=use_ssim not in (True, False, 0, 1, 2, 3),
                           cv_patch_size=cv_patch_size not in (1, 3, 5, 7), augmentation=augmentation not in (None, "none"))
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(
                "monorec_amd implements the default inference configuration of the reference only; "
                f"unsupported non-default options: {bad} (SURVEY.md section 8 f-4)")
        self._hip_graph = bool(hip_graph)
        self._in_flight = max(1, int(hip_in_flight))
        # submit() coalesces this many consecutive equal-shaped requests into ONE launch of the path (dynamic batching of a keyframe
        # stream: at batch 1 two thirds of the launches are latency chains); 1 = every request is launched on its own
        self._batch_keyframes = max(1, int(hip_batch_keyframes))
        self._open_group = None
        # how far the host may run ahead of the GPU: submit() blocks until all but the last `hip_queue_depth - 1` earlier forwards of
        # the slot it is about to reuse have finished, i.e. at most hip_in_flight * hip_queue_depth forwards are ever enqueued.
        # Measured (round 3, 48-run grid tools/sessions/r03_s3.sh): 1 is best - a forward enqueued while its slot is still busy
        # (depth 2 and more) costs 5-7 % keyframes/s, whatever stream the requests come in on.
        self._queue_depth = max(1, int(hip_queue_depth))
        # forward() (not submit()): launch on the caller's current stream instead of the slot's streams (round 6; False = rounds 3-5: slot streams,
        # host-side input wait, wait packet on the caller's stream)
        self._forward_inline = bool(hip_forward_on_callers_stream)
        # profiling aid: encoder and main stages on ONE stream, so that a kernel trace of `hip_in_flight=1` shows isolated kernel
        # durations (with two streams the ResNet launches overlap the cost volume / mask encoder and inflate each other)
        self._single_stream = bool(hip_single_stream)
        # Streams per in-flight slot of submit(): 1 (default since round 5) = all stages of a keyframe on ONE stream, 2 = encoder stage on a second
        # stream (rounds 2-4).  The GPU runs four hardware queues side by side; measured at c2 (tools/sessions/r05_s8.sh, s9.sh, 200 / 20 steps): four
        # slots x one stream 820-827 / 749-754 keyframes/s, two slots x two streams 762-766 / 716-719, three x one 791, four x two 794, five x one
        # 744.  One keyframe at a time - forward(), or hip_in_flight=1 - keeps its encoder stage on a second stream ("e0": 1187 vs 1278 us per
        # keyframe, r04_s17): None = 2 streams with one slot, 1 stream per slot otherwise.
        if hip_slot_streams is None:
            hip_slot_streams = 2 if self._in_flight == 1 else 1
        self._slot_streams_n = 2 if int(hip_slot_streams) >= 2 else 1
        # One-stream-per-slot mode: how many HIP streams the slots share (slot s runs on stream s % hip_streams).  The GPU runs about four hardware queues
        # side by side (a fifth stream costs: 5 x 1 744 against 4 x 1 820-827 keyframes/s), so more than four slots means more than one slot per stream:
        # the second keyframe of a stream is already enqueued when the first finishes.  Measured (tools/sessions/r05_s17.sh, c2, 200 steps): 8 slots on 4
        # streams 830-833 = 4 slots (825-831) - the four streams are not what idles -, on 8 streams 805, on 3 streams 786, 6 slots (uneven) 800; 12 slots at
        # c2 and 8 at the configs[4] shape collapse (141-177 / 110 keyframes/s, the host blocks inside the launch calls).  More than four slots buy nothing.
        # Each slot keeps its own resident buffers, so results stay valid until the slot is reused.
        if self._in_flight > 8:
            warnings.warn(f"monorec_amd: hip_in_flight={self._in_flight}: more than 8 keyframes enqueued at once made the launch calls block on the MI355X "
                          "(141-177 instead of 830 keyframes/s at 256x512, tools/sessions/r05_s17.sh); 4 is the measured optimum")
        self._n_streams = self._in_flight if self._slot_streams_n == 2 else max(1, min(self._in_flight, int(hip_streams) if hip_streams else 4))
        self.host_enqueue_stats = [0, 0.0]   # forwards enqueued, host seconds spent enqueueing them (without the run-ahead waits)
        # convolution arithmetic: 0 fp32 MFMA (default; the 1e-4 parity path), 1 bf16 MFMA (hip_bf16: weights / activations rounded
        # to bf16, fp32 accumulate - BASELINE configs[4], NOT within the parity bar), 2 bf16x3 split (hip_bf16x3, EXPERIMENTAL:
        # hi/lo bf16 pairs, three bf16 MFMAs per product - fp32-class accuracy, 4e-6 in CPU emulation)
        self._bf16 = 2 if hip_bf16x3 else (1 if hip_bf16 else 0)
        # which reduced-multiply convolution forms the measured table may select (engine.Plan conv_forms; INTEGRATION.md "trained
        # checkpoints"): False = all of it; "f2" = F(2,.) forms only; True = none - every convolution is the direct MFMA kernel's
        # exact fmaf chain (the reference's arithmetic up to summation order)
        if hip_exact_convs not in (False, True, "f2"):
            raise ValueError("hip_exact_convs must be False, True or 'f2'")
        self._conv_forms = {False: "table", True: "direct", "f2": "f2"}[hip_exact_convs]
        # opt-in: the cost volume's 3x3 window sums formed separably (mr_cost_volume_relaxed_f32) - the sad kernel runs 17 %
        # fewer instructions, validity is unchanged, the volumes move by <= 1e-4 and the depth by <= 2e-6 (inside the 1e-4-on-depth bar); the
        # default keeps the reference's summation order (volumes within 5e-7 of the reference)
        self._cv_separable = bool(hip_cv_separable)
        # opt-in, bf16 configuration only: the output dict carries no `single_frame_cvs` (nothing outside the model reads them: evaluater.py:87,
        # create_pointcloud.py:70-84 use `result` / `cv_mask`; the MaskModule reads the B8 copies) - their fp32 stores are skipped
        self._lean_outputs = bool(hip_lean_outputs)
        self._skip_layer4 = bool(hip_skip_dead_layer4)
        if self._lean_outputs and self._bf16 != 1:
            raise ValueError("hip_lean_outputs needs hip_bf16=True (the fp32 path hands out `single_frame_cvs` like the reference)")
        self._slot_counter = [0]         # mutable on purpose: nn.DataParallel replicas (shallow copies made per forward) share it
        self._plans = {}
        self._graphs = {}
        self._streams = {}
        self._dev_streams = {}           # device -> every stream of this model there, created in one fixed order (_device_streams)
        self._consts = {}
        self._const_slab = {}
        self._packed_state = None
        self._prep_pinned = {}           # (device, matrices, batch) -> (pinned host buffer, stream) of prepare()'s gather launch
        self._lock = threading.RLock()   # one enqueue at a time per model object (nn.DataParallel calls replicas from threads)
        self._warned_encoder = False
        self._geometry_override = None   # (kinv (B,9), proj (B,F,12)) CPU tensors replacing host_geometry's result; tests only

        self._feature_extractor = ResnetEncoder(num_layers=18, pretrained=True)
        if self.freeze_resnet:
            for p in self._feature_extractor.parameters(True):
                p.requires_grad_(False)
        self.cv_module = CostVolumeModule(use_mono=use_mono, use_stereo=use_stereo, use_ssim=use_ssim,
                                          sfcv_mult_mask=self.sfcv_mult_mask, patch_size=cv_patch_size)
        if self.pretrain_mode not in (1, 3):                                       # :622-626
            if not self.simple_mask:
                self.att_module = MaskModule(self.cv_depth_steps, self._feature_extractor.num_ch_enc,
                                             use_cv=mask_use_cv, use_features=mask_use_feats)
            else:
                self.att_module = SimpleMaskModule(self.cv_depth_steps, self._feature_extractor.num_ch_enc)
        if self.pretrain_mode != 2:                                                # :627-628
            self.depth_module = DepthModule(self.cv_depth_steps, feature_channels=self._feature_extractor.num_ch_enc,
                                            large_model=self.depth_large_model)
        self.augmenter = None

        def load(cp_list, sub=None, prefix=None):
            if cp_list is None:
                return
            for cp in (cp_list if isinstance(cp_list, (list, tuple)) else [cp_list]):
                checkpoint = torch.load(cp, map_location=torch.device("cpu"))
                sd = _filter_state_dict(checkpoint["state_dict"], checkpoint["arch"] == "DataParallel")
                if sub is None:
                    self.load_state_dict(sd, strict=False)
                else:
                    sub.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix)}, strict=False)

        load(self.checkpoint_location)                                             # monorec_model.py:630-637
        if self.mask_cp_loc is not None:
            load(self.mask_cp_loc, self.att_module, "att_module")                  # :639-647
        if self.depth_cp_loc is not None:
            load(self.depth_cp_loc, self.depth_module, "depth_module")             # :649-657
        for module_name in self.freeze_module:                                     # :659-663
            module = getattr(self, module_name + "_module")
            module.eval()
            for param in module.parameters(True):
                param.requires_grad_(False)
        self.register_load_state_dict_post_hook(MonoRecModel._after_load_state_dict)

    @staticmethod
    def _after_load_state_dict(module, incompatible):
        module._invalidate()
        if not any(k.startswith("_feature_extractor.") and not k.endswith("num_batches_tracked") for k in incompatible.missing_keys):
            module._feature_extractor._weights_loaded = True

    # ------------------------------------------------------------------ plan management
    def _invalidate(self):
        self._plans = {}
        self._graphs = {}
        self._consts = {}
        self._const_slab = {}
        self._packed_state = None

    def _weights_snapshot(self):
        """CPU fp32 copy of the weights, shared by the plans of every slot / device (and by nn.DataParallel replicas)."""
        if self._packed_state is None:
            self._packed_state = ("cpu", {k: v.detach().to("cpu", torch.float32) for k, v in self.state_dict().items()})
        return self._packed_state[1]

    def _replicate_for_data_parallel(self):
        # nn.DataParallel replicates the module on every forward; a replica's `_parameters` are empty (torch sets the weight copies
        # as plain attributes), so its state_dict() holds buffers only.  The snapshot is therefore taken here, on the original,
        # and travels to the replicas through the shallow copy of __dict__ - also when the very first forward is a wrapped one
        # (evaluater.py:27-30 wraps right after construction / checkpoint load).
        self._weights_snapshot()
        return super()._replicate_for_data_parallel()

    def _apply(self, fn, *a, **k):   # .to() / .cuda(): parameters moved or cast -> repack
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def __getstate__(self):          # copy.deepcopy / pickle: device plans, streams, graphs and the lock are rebuilt on demand
        state = dict(super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__)
        for k in ("_plans", "_graphs", "_streams", "_dev_streams", "_consts", "_const_slab", "_prep_pinned"):
            state[k] = {}
        state["_packed_state"] = None
        state["_open_group"] = None
        state.pop("_lock", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._lock = threading.RLock()

    @property
    def hip_in_flight(self):
        """Keyframes a stream of submit() calls keeps on the GPU at once (the constructor's `hip_in_flight`)."""
        return self._in_flight

    def _device_streams(self, device):
        """Every HIP stream this model uses on `device`, created AND FIRST USED at one point in a FIXED order: the main streams of the in-flight slots, the
        encoder streams (one per slot with `hip_slot_streams=2`, else only forward()'s "e0"), then the gather stream of prepare().  The order is not cosmetic:
        ROCm binds a stream to a hardware queue when the stream is first USED, in order of first use, and where the model's busy streams sit among the queues
        sets the pipelined rate for the life of the process - measured at c2 with two slots x two streams (tools/sessions/r05_s6.sh, r05_s7.sh, 200 steps):
        the four next to each other, in any order, 757-769 keyframes/s; another stream first used between them 693-717; spread out with unused streams between
        them 509-558.  Round 4 had a good order by accident of its call order (750); the first tree of round 5 lost 8 % when its first request happened to
        launch on an encoder stream before the gather stream (r05_s1 - s5).  The GPU runs about four queues side by side, and four slots with ONE stream each
        use them best (r05_s9: 820-827 against 762-766 for 2 x 2, 794 for 4 x 2, 744 for 5 x 1): the default.  So every stream gets one 4-byte launch here,
        in `_STREAM_LAYOUT` order ("g" gather, "m<slot>" / "e<slot>" main / encoder stream of a slot, "_" an extra stream that only takes a queue);
        MR_DIAG_STREAM_LAYOUT: experiments only."""
        key = str(device)
        if key not in self._dev_streams:
            # one stream per slot: forward() borrows the next slot's stream for its encoder stage (_slot_streams); an "e0" only if there is no next slot
            n_enc = self._in_flight if self._slot_streams_n == 2 else (1 if self._n_streams == 1 else 0)
            layout = _STREAM_LAYOUT or ",".join([f"m{s_}" for s_ in range(self._n_streams)] + [f"e{s_}" for s_ in range(n_enc)] + ["g"])
            names = layout.split(",")
            names += [n_ for n_ in [f"m{s_}" for s_ in range(self._n_streams)] + [f"e{s_}" for s_ in range(n_enc)] + ["g"] if n_ not in names]   # left out: behind
            made, pads = {}, []
            touch = torch.zeros(len(names), dtype=torch.float32, device=device)
            torch.cuda.synchronize(device)
            for i, name in enumerate(names):
                st = torch.cuda.Stream(device, priority=-1 if (name[:1] in _STREAM_PRIO and name != "_") else 0)
                with torch.cuda.stream(st):
                    touch[i:i + 1].fill_(1.0)            # the stream's first launch: this is when it gets its hardware queue
                st.synchronize()
                if name == "_":
                    pads.append(st)
                else:
                    made[name] = st
            made["_pads"] = pads                         # (kept alive)
            self._dev_streams[key] = made
        return self._dev_streams[key]

    def _slot_streams(self, slot, device, own=False):
        """{"main", "enc"} of an in-flight slot.  `own` = forward(): with one stream per slot the encoder stage of a forward() still runs beside the cost volume -
        on the NEXT slot's stream (its hardware queue is a neighbour of this slot's; a fifth stream behind the four slots shares a hardware pipe with the
        first: forward() 566 -> 533 keyframes/s, r05_s10).  forward() calls are sequential; a submit() pending on that slot only delays the stage (stream order)."""
        st = self._streams.get((slot, str(device), bool(own)))
        if st is None:
            ds = self._device_streams(device)
            main = ds[f"m{slot % self._n_streams}"]
            if self._single_stream:
                enc = main
            elif self._slot_streams_n == 2:
                enc = ds[f"e{slot}"]
            else:
                enc = main if not own else (ds[f"m{(slot + 1) % self._n_streams}"] if self._n_streams > 1 else ds["e0"])
            st = {"main": main, "enc": enc}
            self._streams[(slot, str(device), bool(own))] = st
        return st

    def _plan_for(self, slot, batch, h, w, nf, device):
        key = (slot, batch, h, w, nf, self.cv_depth_steps, str(device))
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(self._weights_snapshot(), batch, h, w, nf, self.cv_depth_steps, self.inv_depth_min_max, device,
                        alpha=self.cv_module.alpha, channel_weights=self.cv_module.channel_weights, bf16=self._bf16,
                        use_ssim=self.use_ssim, sfcv_mult_mask=self.sfcv_mult_mask, pretrain_mode=self.pretrain_mode,
                        no_cv=self.no_cv, mask_use_cv=self.mask_use_cv or self.simple_mask,
                        mask_use_feats=self.mask_use_feats or self.simple_mask, simple_mask=self.simple_mask,
                        cv_patch_size=self.cv_patch_size, conv_forms=self._conv_forms, cv_separable=self._cv_separable, lean_outputs=self._lean_outputs, skip_layer4=self._skip_layer4)
            plan.buf["depths"].copy_(depth_hypotheses(self.inv_depth_min_max, self.cv_depth_steps))
            plan.host_geom = torch.empty(batch * 9 + batch * nf * 12, dtype=torch.float32).pin_memory()
            plan.host_mats = torch.empty(2 + 2 * nf, batch, 4, 4, dtype=torch.float32).pin_memory()
            plan.geom_uploaded = None       # event behind the last H2D copy out of host_geom
            plan.host_time = torch.zeros(8, dtype=torch.float32).pin_memory()   # ring: one scalar per submit, reused 8 submits of this slot later
            plan.host_time_at = 0
            plan.enqueued = collections.deque()   # completion events of this slot's forwards the host has not waited for yet
            plan.consumers = []                   # streams that were handed results of this slot since its last enqueue
            plan.handles = []                     # weak references to the submit() handles whose outputs are views of this slot
            plan.own_layout = None                # forward(): arena layout of the caller-owned outputs (see _bind_owned_outputs)
            self._plans[key] = plan
        return key, plan

    # ------------------------------------------------------------------ forward
    # tensor-valued outputs the path writes into the dict (monorec_model.py:256-279,690,713-727)
    _OUTPUT_KEYS = ("cost_volume", "single_frame_cvs", "image_features", "cv_mask", "predicted_inverse_depths",
                    "inv_depth_min", "inv_depth_max", "cv_depth_steps")

    def forward(self, data_dict):
        """Reference contract (monorec_model.py:672-729): fills and returns `data_dict`; the outputs are ordered on the
        caller's current stream like any PyTorch op and - like the reference's - are tensors the caller OWNS: they stay
        valid whatever is run afterwards (create_pointcloud.py:79-98 keeps `result` of five keyframes and multiplies the
        middle one in place).  `submit()` is the zero-copy interface."""
        # the lock is held until the copies are enqueued: nn.DataParallel replicas of one device (threads sharing the per-device
        # plans) may land on the same slot, and the next enqueue on a slot overwrites the buffers these copies read - enqueued
        # under the lock, the copies sit on the caller's stream in front of the event the next forward's launches wait for
        with self._lock:
            # stream order makes consecutive forward() calls sequential whatever the number of slots, so they all use slot 0: one
            # set of resident buffers and packed weights stays hot (alternating two slots measured 2.1 -> 2.8 ms per forward)
            self._flush_open_group()
            handle = self._forward_handle(data_dict)
            out = handle.result()
            if not handle.owned:                              # plan variants with constant content in their output buffers, hipGraph
                with torch.cuda.device(out["keyframe"].device):
                    self._own_outputs(out)
        if self.pretrain_mode == 2:                           # :723-727, same aliasing as the reference
            out["result"] = out["cv_mask"]
        else:
            out["result"] = out["predicted_inverse_depths"][0]
            out["mask"] = out["cv_mask"]
        return out

    def _forward_handle(self, data_dict):
        """Enqueue forward()'s keyframe: checks, host wait for the inputs, then the launches - with the outputs bound to memory the
        caller will own where the plan allows it (`handle.owned`)."""
        data_dict.pop(_METRIC_CACHE_KEY, None)
        prep = self._parse(data_dict)                         # checks only; the pose algebra runs behind the encoder's launches
        with torch.cuda.device(prep.device):
            slot = self._forward_slot(prep)
            if self._forward_inline and not self._hip_graph:
                b, h, w, nf = prep.shape
                self._consts_for(prep.device)
                _, plan = self._plan_for(slot, b, h, w, nf, prep.device)
                if plan.outputs_rebindable:
                    # Round 6: the keyframe is enqueued ON THE CALLER'S STREAM.  Stream order then says everything the host used to
                    # wait for - inputs ready, the previous forward done with the slot's buffers, outputs ordered for the caller - so the host waits
                    # for nothing before it launches: the gather of the 4x4s and the pose-independent encoder stage of THIS call queue up behind the
                    # previous call's launches while those still run, and the device never idles between two forwards (the slot-stream path lost
                    # ~0.09 ms per call to the signalling slot stream -> caller's stream -> host -> first launch).
                    return self._submit_locked(data_dict, prep, slot=slot, own=True, inline=True)
            # Everything that needs neither the inputs nor an idle slot happens BEFORE the host waits for the caller's stream (which, in a
            # loop of forward() calls, is the wait for the previous forward): the two output arenas and the re-targeting of the plan's
            # launches (~50 us).  The previous forward's launches copied their descriptors when they were enqueued, so patching them now
            # is safe; behind the wait only the launches themselves remain (forward() at 0.92 x the in-flight-1 loop).
            prebound = self._prebind_owned(prep, slot)
            self._wait_inputs(prep.device)
            return self._submit_locked(data_dict, prep, slot=slot, own=True, prebound=prebound)

    def _consts_for(self, device):
        """The three constants of monorec_model.py:675-677, built once per device (a `new_tensor` from a Python list is a blocking pageable
        H2D copy on the caller's stream, three of them per keyframe) as one 16-byte slab; forward() hands out a copy with the other outputs."""
        consts = self._consts.get(str(device))
        if consts is None:
            slab = torch.zeros(4, dtype=torch.float32, device=device)
            slab[0], slab[1] = self.inv_depth_min_max[0], self.inv_depth_min_max[1]
            slab.view(torch.int32)[2] = int(self.cv_depth_steps)
            consts = (slab[0:1], slab[1:2], slab.view(torch.int32)[2:3])
            self._consts[str(device)] = consts
            self._const_slab[str(device)] = slab
        return consts

    def _prebind_owned(self, prep, slot):
        """forward(): allocate the caller-owned output arenas and point the slot's plan at them, ahead of the input wait.  None when the
        plan hands out copies instead (constant content in its output buffers, hipGraph replay).  Safe while an earlier forward of the
        slot is still running: every launch copies its descriptor when it is enqueued, and nothing of this model is enqueued lazily."""
        if self._hip_graph:
            return None
        b, h, w, nf = prep.shape
        self._consts_for(prep.device)
        _, plan = self._plan_for(slot, b, h, w, nf, prep.device)
        if not plan.outputs_rebindable:
            return None
        streams = self._slot_streams(slot, prep.device, own=True)
        return self._bind_owned_outputs(plan, prep.device, (streams["main"], streams["enc"]))

    def _own_outputs(self, out):
        """Replace the output views of `out` (resident slot buffers) by tensors the caller owns: one allocation, ONE copy launch
        (mr_copy_segments) on the caller's stream instead of one ATen clone per tensor (~16 launches at c2)."""
        device = out["keyframe"].device
        todo = []                                             # (key, list index or None, tensor)
        for k in self._OUTPUT_KEYS:
            v = out.get(k)
            if torch.is_tensor(v):
                todo.append((k, None, v))
            elif isinstance(v, list):
                out[k] = list(v)
                todo += [(k, i, t) for i, t in enumerate(v)]
        fast, offset = [], 0
        ckeys = ("inv_depth_min", "inv_depth_max", "cv_depth_steps")
        cvals = [out.get(k) for k in ckeys]
        consts = None
        if all(torch.is_tensor(c) and c.is_cuda for c in cvals) and cvals[1].data_ptr() == cvals[0].data_ptr() + 4 and \
                cvals[2].data_ptr() == cvals[0].data_ptr() + 8 and cvals[0].data_ptr() % 16 == 0:
            consts, offset = cvals[0].data_ptr(), 256         # the three constants of :675-677 travel as one 16-byte segment
            todo = [e for e in todo if e[0] not in ckeys]
        for k, i, t in todo:
            nbytes = t.numel() * t.element_size()
            if t.is_cuda and t.is_contiguous() and nbytes >= 16 and nbytes % 16 == 0 and t.data_ptr() % 16 == 0 and t.device == device:
                fast.append((k, i, t, offset, nbytes))
                offset += (nbytes + 255) // 256 * 256
            else:                                             # the three one-element constants, anything unusual
                c = t.clone()
                if i is None:
                    out[k] = c
                else:
                    out[k][i] = c
        if not fast and consts is None:
            return
        arena = torch.empty(offset, dtype=torch.uint8, device=device)
        base = arena.data_ptr()
        if consts is not None:
            fast.insert(0, (None, None, None, 0, 16))
            out["inv_depth_min"], out["inv_depth_max"] = arena[0:4].view(torch.float32), arena[4:8].view(torch.float32)
            out["cv_depth_steps"] = arena[8:12].view(torch.int32)
        stream = torch.cuda.current_stream(device).cuda_stream
        lib = _lib.load()
        for lo in range(0, len(fast), _lib.MR_MAX_COPY_SEGMENTS):
            part = fast[lo:lo + _lib.MR_MAX_COPY_SEGMENTS]
            segs = (_lib.CopySegment * len(part))()
            for j, (_, _, t, off, nbytes) in enumerate(part):
                segs[j].src, segs[j].dst, segs[j].bytes = (consts if t is None else t.data_ptr()), base + off, nbytes
            _lib.check(lib.mr_copy_segments(segs, len(part), stream), "mr_copy_segments")
        for k, i, t, off, nbytes in fast:
            if t is None:
                continue
            c = arena[off:off + nbytes].view(t.dtype).view(t.shape)
            if i is None:
                out[k] = c
            else:
                out[k][i] = c

    def prepare(self, data_dict):
        """Optional first half of `submit()`: everything a forward needs from its inputs but not from a free slot - the checks and the
        host-side pose algebra (with the 4x4s on the device: one gather launch into pinned host memory, awaited here).  A pipelined
        loop calls it BEFORE it waits for the result whose slot the next submit reuses:

            token = model.prepare(batch);  out = pending.popleft().synchronize();  pending.append(model.submit(batch, token))

        so that ~0.1-0.2 ms of host work per keyframe overlap the device instead of preceding the first launch of the keyframe.
        Returns a token for `submit(data_dict, token)`; the inputs must not change in between."""
        prep = self._parse(data_dict)
        with self._lock, torch.cuda.device(prep.device):
            self._wait_inputs(prep.device)
            # (A parse-only token for the first request on an idle device - pose algebra behind the encoder stage's launches, as forward() does it - was
            # built and measured in round 5: 705-709 against 716-718 keyframes/s on 20-step lines, r05_s6; again in round 6 with one stream per slot:
            # 764-783 against 760-772, r06_s30 - the cold first round trip is paid inside submit() instead.  Not kept.)
            self._geometry(prep)
        return prep

    def _parse(self, data_dict):
        """The checks of a request and its tensors, without touching the device: a `_Prepared` whose matrices are still to be formed."""
        if self.training:
            raise NotImplementedError("monorec_amd.MonoRecModel is inference-only: call .eval() first")
        keyframe = data_dict["keyframe"]                      # missing keys -> KeyError, like the reference
        kf_intrinsics, kf_pose = data_dict["keyframe_intrinsics"], data_dict["keyframe_pose"]
        frames, poses, intrinsics = [], [], []
        if self.no_cv:                                        # :682-686: one zero volume per entry of data_dict["poses"]
            frames, intrinsics, poses = list(data_dict["frames"]), list(data_dict["intrinsics"]), list(data_dict["poses"])
        elif self.use_mono:                                   # monorec_model.py:160-163
            frames += list(data_dict["frames"])
            intrinsics += list(data_dict["intrinsics"])
            poses += list(data_dict["poses"])
        if self.use_stereo and not self.no_cv:                # :164-167: the stereo frame is one more source view
            frames += [data_dict["stereoframe"]]
            intrinsics += [data_dict["stereoframe_intrinsics"]]
            poses += [data_dict["stereoframe_pose"]]
        if not keyframe.is_cuda:
            raise RuntimeError("monorec_amd.MonoRecModel needs its inputs on a HIP device (cuda:N on ROCm); "
                               "there is no CPU path")
        _lib.load()
        if not self._feature_extractor._weights_loaded and not self._warned_encoder:
            self._warned_encoder = True
            warnings.warn("monorec_amd.MonoRecModel: the ResNet-18 encoder has neither ImageNet weights (torchvision is not "
                          "importable / has no cached weights; the reference builds resnet18(pretrained=True), "
                          "monorec_model.py:104-113) nor `_feature_extractor.*` entries from a checkpoint or state dict: "
                          "it runs with its random initialisation")
        cv_depths = data_dict.get("cv_depths")                # per-pixel depth hypotheses (monorec_model.py:181-182)
        if cv_depths is not None and self._hip_graph:
            raise NotImplementedError("cv_depths with hip_graph=True: the captured launch has no per-pixel depth pointer")
        b, c, h, w = keyframe.shape
        nf = len(frames)
        mat_list = [kf_intrinsics, kf_pose] + intrinsics + poses
        for m in mat_list:                                    # the gather launch below reads 16 * b floats per matrix: no silent overrun
            if tuple(m.shape) != (b, 4, 4):
                raise ValueError(f"pose / intrinsics matrices must be ({b}, 4, 4) like the keyframe batch, got {tuple(m.shape)}")
        return _Prepared(data_dict, keyframe, frames, cv_depths, (b, h, w, nf), keyframe.device, None, None, mat_list)

    def _wait_inputs(self, device):
        """The HOST waits for the caller's stream to reach this point: the inputs exist now, and nothing afterwards (here or in
        submit) is enqueued behind an unsatisfied stream dependency - a launch parked in its hardware queue as a blocked barrier
        packet slows the OTHER queues down (measured, round 3: sequential forwards 2.1 -> 3.2 ms with the next forward's encoder
        pre-enqueued behind such a wait; a request stream enqueued one forward ahead 5-7 % slower)."""
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(torch.cuda.current_stream(device))
        _host_wait(inputs_ready)

    def _geometry_begin(self, prep, stream=None):
        """First half of the pose algebra of a parsed request: with the 4x4s on the device, the ONE gather launch that brings them into
        device-writable pinned host memory (its own stream; nothing waits here).  Returns the state `_geometry_finish` completes.
        forward() / an idle-device submit() call this BEFORE they enqueue the encoder stage, so that the round trip of the gather runs
        while the host enqueues those ~25 launches instead of queueing behind them."""
        device, mat_list = prep.device, prep.mats
        b, _, _, nf = prep.shape
        caller = torch.cuda.current_stream(device)
        if all(not m.is_cuda for m in mat_list):
            # matrices on the host (a loader that keeps the 4x4s on the CPU, kitti.KittiOdometryDataset): used where they are
            return [m.detach().float() for m in mat_list], None
        # matrices on the device: one gather launch into device-writable pinned host memory
        dm = [m if (m.dtype == torch.float32 and m.is_contiguous() and m.device == device) else
              m.to(device=device, dtype=torch.float32).contiguous() for m in mat_list]
        if stream is None and any(x is not y for x, y in zip(dm, mat_list)):     # a conversion ran on the caller's stream: let it finish first
            ev = torch.cuda.Event()
            ev.record(caller)
            _host_wait(ev)
        pk = (str(device), len(dm), b)
        pinned = self._prep_pinned.get(pk)
        if pinned is None:
            pinned = self._prep_pinned[pk] = (torch.empty(len(dm), b, 4, 4, dtype=torch.float32).pin_memory(), self._device_streams(device)["g"])
        hm, gs = pinned
        if stream is not None:             # forward() on the caller's stream: the gather is ordered behind the matrices by stream order
            gs = stream
        ptrs = (ctypes.c_void_p * len(dm))(*[m.data_ptr() for m in dm])
        _lib.check(_lib.load().mr_gather_small_f32(ptrs, len(dm), 16 * b, hm.data_ptr(), gs.cuda_stream), "mr_gather_small_f32")
        done = torch.cuda.Event()
        done.record(gs)
        for m in dm:
            m.record_stream(gs)
        return hm, done

    def _geometry_finish(self, prep, state):
        """Second half: wait for the gather, then kinv / proj (monorec_model.py:171,198,207; layers.py:65) into the token."""
        hm, done = state
        b, _, _, nf = prep.shape
        if done is not None:
            _host_wait(done)
        # host 4x4 algebra with the same ATen CPU operators as the reference: bit-identical matrices (~0.1 ms)
        kinv, proj = host_geometry(hm[0], hm[1], [hm[2 + f] for f in range(nf)], [hm[2 + nf + f] for f in range(nf)])
        if self._geometry_override is not None:           # test seam: matrices formed on another host (tests/golden geom.*)
            kinv, proj = self._geometry_override
        prep.kinv, prep.proj = kinv.reshape(-1).clone(), proj.reshape(-1).clone()
        return prep

    def _geometry(self, prep):
        """kinv / proj of a parsed request into the token.  Inputs must be ready."""
        return self._geometry_finish(prep, self._geometry_begin(prep))

    def _forward_slot(self, prep):
        """Slot of a forward() call: 0 - one set of resident buffers and packed weights stays hot -, unless its plan hands out copies of
        its resident output buffers (no rebinding) and a submit() handle of that slot has not been collected yet: the copies' source
        would be that handle's outputs.  Then the next slot without one."""
        b, h, w, nf = prep.shape
        for slot in range(self._in_flight):
            plan = self._plans.get((slot, b, h, w, nf, self.cv_depth_steps, str(prep.device)))
            if plan is None or (plan.outputs_rebindable and not self._hip_graph):
                return slot
            if not any(hd() is not None and not hd().collected for hd in plan.handles):
                return slot
        raise RuntimeError("forward(): every in-flight slot holds a submit() whose result has not been taken; collect one "
                           "(handle.result() / .synchronize()) first, or construct the model with a larger hip_in_flight")

    def _submit_one(self, data_dict, prepared=None, slot=None):
        """Enqueue one forward on the next in-flight slot and return a handle without making the caller's stream
        wait for it.  Keyframes are independent, so a stream of keyframes is served with `hip_in_flight` (default 4)
        of them on the GPU at once - on separate HIP streams and resident buffers - which fills the launch
        head/tail bubbles that a single batch-1 keyframe leaves on 256 CUs:

            if len(pending) == hip_in_flight: out = pending.popleft().synchronize()
            pending.append(model.submit(batch))

        The inputs must stay unmodified until the result is taken.  Zero-copy: the outputs are VIEWS of the slot's resident buffers
        and are overwritten by the submit that reuses the slot (`hip_in_flight` submits later) - consume or clone them
        before that.  `forward()` returns owned tensors instead."""
        if prepared is None:
            prepared = self.prepare(data_dict)
        elif prepared.data is not data_dict:
            raise ValueError("submit(data_dict, prepared): the token was prepared for another dict")
        data_dict.pop(_METRIC_CACHE_KEY, None)                # cached metric sums of an earlier forward on this dict are stale now
        with self._lock, torch.cuda.device(prepared.device):
            return self._submit_locked(data_dict, prepared, slot)

    # keys of a request that dynamic batching concatenates along the batch dimension (tensors, or lists of tensors)
    _BATCHED_INPUTS = ("keyframe", "keyframe_intrinsics", "keyframe_pose", "frames", "intrinsics", "poses",
                       "stereoframe", "stereoframe_intrinsics", "stereoframe_pose")

    def submit(self, data_dict, prepared=None):
        """Enqueue one forward and return a handle (`.result()` -> the filled dict, outputs are views: see `_submit_one`).

        With `hip_batch_keyframes = K > 1`, K consecutive requests of equal shapes are coalesced into one launch of the path over
        their concatenated batch (keyframes are independent): the launch goes out when the K-th request arrives, or when
        the result of a member is asked for earlier (then with the members collected so far).  Requests carrying per-request
        extras (`cv_depths`, `mvobj_mask` with pretrain_mode 3, `simple_mask`'s previous prediction) are launched on their own."""
        if self._batch_keyframes <= 1 or self.simple_mask or self.pretrain_mode == 3 or data_dict.get("cv_depths") is not None:
            self._flush_open_group()
            return self._submit_one(data_dict, prepared)
        with self._lock:
            data_dict.pop(_METRIC_CACHE_KEY, None)            # cached metric sums of an earlier forward on this dict are stale now
            # a request is checked when it is submitted, not when some later call happens to launch its group
            if self.training:
                raise NotImplementedError("monorec_amd.MonoRecModel is inference-only: call .eval() first")
            kf = data_dict["keyframe"]                        # missing keys -> KeyError, like the reference
            for k in self._required_inputs():
                data_dict[k]
            sig = (tuple(kf.shape), str(kf.device), len(data_dict.get("frames", ())))
            g = self._open_group
            if g is not None and g.sig != sig:
                self._flush_open_group()
                g = None
            if g is None:
                g = self._open_group = _Group(sig)
            g.members.append(data_dict)
            handle = _GroupHandle(self, g, len(g.members) - 1)
            if len(g.members) >= self._batch_keyframes:
                self._flush_open_group()
            return handle

    def _required_inputs(self):
        """Input keys `_submit_one` reads for the current options (monorec_model.py:160-167,672-686)."""
        keys = ["keyframe", "keyframe_intrinsics", "keyframe_pose"]
        if self.no_cv or self.use_mono:
            keys += ["frames", "intrinsics", "poses"]
        if self.use_stereo and not self.no_cv:
            keys += ["stereoframe", "stereoframe_intrinsics", "stereoframe_pose"]
        return keys

    def _flush_open_group(self):
        """Launch the requests collected so far as one batch and hand every member its slice of the outputs.  A launch that fails
        leaves its exception on the group: every member's handle re-raises it instead of tripping over a missing launch."""
        with self._lock:
            g, self._open_group = self._open_group, None
            if g is None or g.pending is not None or g.error is not None:
                return
            try:
                self._launch_group(g)
            except BaseException as e:
                g.error = e
                raise

    def _launch_group(self, g):
        ms = g.members
        if len(ms) == 1:
            g.pending = self._submit_one(ms[0])
            return
        combined = {}
        for k in self._BATCHED_INPUTS:
            if k not in ms[0]:
                continue
            v = ms[0][k]
            combined[k] = ([torch.cat([m[k][i] for m in ms]) for i in range(len(v))] if isinstance(v, (list, tuple))
                           else torch.cat([m[k] for m in ms]))
        g.pending = self._submit_one(combined)
        b = ms[0]["keyframe"].shape[0]
        for j, m in enumerate(ms):
            lo, hi = j * b, (j + 1) * b
            for k, v in combined.items():
                if k in self._BATCHED_INPUTS:
                    continue
                if torch.is_tensor(v):
                    m[k] = v[lo:hi] if (v.dim() == 4 and v.shape[0] == b * len(ms)) else v
                elif isinstance(v, list):
                    m[k] = [t[lo:hi] for t in v]
            if self.pretrain_mode == 2:
                m["result"] = m["cv_mask"]
            else:
                m["result"] = m["predicted_inverse_depths"][0]
                m["mask"] = m["cv_mask"]

    def _submit_locked(self, data_dict, prepared, slot=None, own=False, prebound=None, inline=False):
        """`own`: forward() - the outputs are produced in memory the caller owns (two arenas allocated here, every launch that writes
        or reads an output buffer re-targeted by Plan.rebind_outputs) instead of the slot's resident buffers; a token without
        matrices gets them formed behind the encoder's launches (the device starts on the pose-independent stage while the host
        does the pose algebra - the caller of forward() has nothing else to overlap it with)."""
        keyframe, frames, cv_depths, device = prepared.keyframe, prepared.frames, prepared.cv_depths, prepared.device
        b, h, w, nf = prepared.shape
        # the three constants of :675-677: built once per device (a `new_tensor` from a Python list is a blocking pageable H2D copy on
        # the caller's stream, three of them per keyframe); forward() hands out copies like every other output
        consts = self._consts_for(device)
        data_dict["inv_depth_min"], data_dict["inv_depth_max"], data_dict["cv_depth_steps"] = consts

        if slot is None:
            slot = self._slot_counter[0]
            self._slot_counter[0] = (slot + 1) % self._in_flight
        key, plan = self._plan_for(slot, b, h, w, nf, device)
        caller = torch.cuda.current_stream(device)
        if inline:                                           # forward(): every launch on the caller's stream (see _forward_handle)
            main = enc = caller
        else:
            streams = self._slot_streams(slot, device, own=own)
            main, enc = streams["main"], streams["enc"]
        # forwards of this slot enqueued on ANOTHER stream (submit() and forward() mixed on one slot): no stream order between them and the
        # launches below, which overwrite the slot's resident buffers - the host waits for them
        for ev, st in [e for e in plan.enqueued if e[1] != main]:
            _host_wait(ev)
        if any(e[1] != main for e in plan.enqueued):
            plan.enqueued = collections.deque(e for e in plan.enqueued if e[1] == main)
        # host run-ahead: at most `hip_queue_depth` forwards of a slot are enqueued at any time (inline: one more - the next call's gather and
        # encoder stage are MEANT to queue up behind the running forward; the wait for the gather bounds the run-ahead by itself)
        while len(plan.enqueued) >= (max(2, self._queue_depth) if inline else self._queue_depth):
            _host_wait(plan.enqueued.popleft()[0])
        start_time = time.time()
        # The launches below overwrite the slot's previous outputs: the host waits for whatever the caller's stream - and any other
        # stream that took results of this slot through `.result()` - has been given to do with them so far (host waits, not stream
        # waits: no blocked packets, see prepare(); an idle stream costs a few microseconds).
        for cs in ([] if own else [caller]) + [c for c in plan.consumers if c != caller]:   # (forward() has just waited for its caller's stream)
            ev = torch.cuda.Event()
            ev.record(cs)
            _host_wait(ev)
        plan.consumers.clear()
        owned = prebound                                     # forward(): arenas allocated and launches re-targeted ahead of the input wait
        if owned is None and own and plan.outputs_rebindable and not self._hip_graph:
            owned = self._bind_owned_outputs(plan, device, (main, enc))
        elif owned is None and not self._hip_graph:
            plan.rebind_outputs(None)
        with torch.cuda.stream(main):
            # the launches read dense fp32 inputs where they are (no device copy); anything else - and hipGraph replay, whose
            # captured launches keep their pointers - goes through the slot's resident buffers
            plan.bind_inputs(keyframe, frames, in_place=not self._hip_graph)
            one = enc is main              # one stream per slot (the default of submit()): stream order is the only dependency, no events
            for t in [keyframe] + frames:
                t.record_stream(main)
                if not one:
                    t.record_stream(enc)
            if not one:
                enc.wait_stream(main)      # behind the inputs and behind the slot's previous main stage (it reads the image features)

            def encoder_stage():
                self._run_stage(key, plan, "encoder", enc)    # ResNet up to layer3, pose independent (its own stream when the slot has two)
                enc_done = None
                if not one:
                    enc_done = torch.cuda.Event()
                    enc_done.record(enc)
                # ResNet layer4 is output-only (image_features[4]): with two streams it keeps running on the encoder stream while the mask /
                # depth stages proceed, and only the completion event of the keyframe waits for it
                self._run_stage(key, plan, "encoder_tail", enc)
                tail_done = None
                if not one:
                    tail_done = torch.cuda.Event()
                    tail_done.record(enc)
                return enc_done, tail_done

            def cv_stage(kinv, proj):
                # one H2D copy of 9 + 12 F floats per sample, then cost volume + mask encoder (concurrent with the ResNet stage)
                if plan.geom_uploaded is not None:
                    _host_wait(plan.geom_uploaded)            # the slot's previous upload has left the pinned buffer (long ago)
                plan.host_geom[: b * 9] = kinv
                plan.host_geom[b * 9:] = proj
                plan.buf["geom"].copy_(plan.host_geom, non_blocking=True)
                plan.geom_uploaded = torch.cuda.Event()
                plan.geom_uploaded.record(main)
                plan.pix_depths_on = cv_depths is not None
                if cv_depths is not None:
                    if tuple(cv_depths.shape) != (b, self.cv_depth_steps, h, w):
                        raise ValueError(f"cv_depths must be (B, cv_depth_steps, H, W), got {tuple(cv_depths.shape)}")
                    if "pix_depths" not in plan.buf:
                        plan.alloc("pix_depths", b, self.cv_depth_steps, h, w)
                    plan.buf["pix_depths"].copy_(cv_depths)
                if self.simple_mask and self.pretrain_mode in (0, 2):
                    # SimpleMaskModule reads a previous prediction from the dict (:453) - KeyError without one, like the reference
                    plan.buf["prev_depth"].copy_(data_dict["predicted_inverse_depths"][0])
                if self.pretrain_mode == 3:                   # :711 cv_mask = data_dict["mvobj_mask"].clone()
                    plan.buf["cv_mask"].copy_(data_dict["mvobj_mask"])
                self._run_stage(key, plan, "cv", main)

            if prepared.kinv is None:      # forward() / idle-device submit(): encoder launches first, the pose algebra while the device runs them
                gstate = self._geometry_begin(prepared, stream=main if inline else None)   # the gather launch goes out ahead of the encoder's ~25 launches ...
                enc_done, tail_done = encoder_stage()
                self._geometry_finish(prepared, gstate)       # ... and has long landed when the host gets here
                cv_stage(prepared.kinv, prepared.proj)
            else:
                cv_stage(prepared.kinv, prepared.proj)   # head of the longest chain (cost volume -> mask encoder -> mask decoder -> depth): first
                enc_done, tail_done = encoder_stage()
            # join: mask decoder -> depth
            if not one:
                main.wait_event(enc_done)
            self._run_stage(key, plan, "main", main)
            if not one:
                main.wait_event(tail_done)
            done = torch.cuda.Event()
            done.record(main)
            plan.enqueued.append((done, main))
        self.host_enqueue_stats[0] += 1
        self.host_enqueue_stats[1] += time.time() - start_time
        # (:279: host seconds spent in the cost-volume module; here: enqueueing.)  The async H2D copy below reads pinned memory some
        # time later: every submit of the slot gets its own scalar of a ring, rewritten only 8 submits of this slot later
        i = plan.host_time_at = (plan.host_time_at + 1) % plan.host_time.numel()
        plan.host_time[i] = time.time() - start_time
        data_dict["cv_module_time"] = plan.host_time[i:i + 1].to(device, non_blocking=True)

        if owned is not None:
            # every launch has copied its descriptor by now: point the plan back at its resident buffers, so that anything that drives
            # the plan directly afterwards (bench.time_layers, tools/*) cannot write into arenas the caller may already have freed
            # (ADVICE r4); off the critical path - the device is busy with this keyframe
            plan.rebind_outputs(None)
            data_dict["cost_volume"] = owned["cost_volume"]
            if "sfcv" in owned:
                data_dict["single_frame_cvs"] = [owned["sfcv"][f] for f in range(nf)]
            data_dict["image_features"] = [owned[f"feat{i}"] for i in range(5) if f"feat{i}" in owned]
            data_dict["cv_mask"] = owned["cv_mask"]
            data_dict["inv_depth_min"], data_dict["inv_depth_max"], data_dict["cv_depth_steps"] = owned["consts"]
            data_dict["predicted_inverse_depths"] = [owned[f"pred{i}"] for i in range(4)]
            data_dict["result"] = data_dict["predicted_inverse_depths"][0]
            data_dict["mask"] = data_dict["cv_mask"]
            return _Pending(data_dict, done, device, None, owned=True, stream=main if inline else None)
        data_dict["cost_volume"] = plan.buf["cost_volume"]
        if not (plan.lean_outputs and plan.b8):               # (lean: the buffer holds raw per-frame costs, not monorec_model.py:251's volumes)
            data_dict["single_frame_cvs"] = [plan.buf["sfcv"][f] for f in range(nf)]
        data_dict["image_features"] = list(plan.feats)
        data_dict["cv_mask"] = plan.buf["cv_mask"]
        if self.pretrain_mode == 2:                           # :723-724: mask only
            data_dict["result"] = data_dict["cv_mask"]
        else:
            data_dict["predicted_inverse_depths"] = list(plan.preds)
            data_dict["result"] = data_dict["predicted_inverse_depths"][0]
            data_dict["mask"] = data_dict["cv_mask"]
        handle = _Pending(data_dict, done, device, plan.consumers)
        if not own:                                           # forward() collects its handle before anyone else can see the slot
            plan.handles = [hd for hd in plan.handles if hd() is not None and not hd().collected]
            plan.handles.append(weakref.ref(handle))
        return handle

    def _bind_owned_outputs(self, plan, device, streams):
        """forward(): allocate the memory the caller will own and re-target the plan's launches at it.  Two arenas - the maps a caller
        typically keeps (`cv_mask`, the four depth scales, the three constants of :675-677: ~1.4 MB at c2) and the bulky ones (fused and
        single-frame volumes, image features: ~62 MB) - so that holding on to `result` does not pin the volumes.  Returns
        name -> tensor views with the resident buffers' shapes."""
        lay = plan.own_layout
        if lay is None:
            lay = {"small": [], "big": [], "size": {"small": 256, "big": 0}}      # the first 256 bytes of the small arena: the constants
            for name in plan.bound:
                if name == "sfcv" and plan.lean_outputs and plan.b8:
                    continue                                  # scratch of the cost-volume kernels, not an output: stays in the resident buffer
                t = plan.buf[name]
                kind = "small" if (name == "cv_mask" or name.startswith("pred")) else "big"
                lay[kind].append((name, lay["size"][kind], tuple(t.shape)))
                lay["size"][kind] += (t.numel() * 4 + 255) // 256 * 256
            plan.own_layout = lay
        small = torch.empty(lay["size"]["small"], dtype=torch.uint8, device=device)
        big = torch.empty(lay["size"]["big"], dtype=torch.uint8, device=device)
        for st in streams:                                    # the arenas are written on the slot's streams, not the allocating one
            small.record_stream(st)
            big.record_stream(st)
        out, bases = {}, {}
        for kind, arena in (("small", small), ("big", big)):
            base = arena.data_ptr()
            for name, off, shape in lay[kind]:
                n = 4
                for d in shape:
                    n *= d
                out[name] = arena[off:off + n].view(torch.float32).view(shape)
                bases[name] = base + off
        plan.rebind_outputs(bases)
        # the three constants: copied from the per-device slab by the caller's stream (16 bytes; ordered like every other output)
        head = small[0:16]
        head.view(torch.float32).copy_(self._const_slab[str(device)], non_blocking=True)
        out["consts"] = (head[0:4].view(torch.float32), head[4:8].view(torch.float32), head[8:12].view(torch.int32))
        return out

    def _run_stage(self, key, plan, stage, stream):
        """Run one stage of the plan on `stream`: eagerly, or (hip_graph) as a captured hipGraph replay."""
        if not plan.stages[stage]:                            # (encoder_tail under hip_skip_dead_layer4: nothing to launch or capture)
            return
        if not self._hip_graph:
            plan.run_stage(stage, stream.cuda_stream)
            return
        gkey = (key, stage)
        entry = self._graphs.get(gkey)
        if entry is None:
            # first call: eager (also sets per-kernel attributes); second call: capture; then replay
            plan.run_stage(stage, stream.cuda_stream)
            self._graphs[gkey] = "warm"
            return
        if entry == "warm":
            graph = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(stream.device)
            cap.wait_stream(stream)
            with torch.cuda.graph(graph, stream=cap):
                plan.run_stage(stage, torch.cuda.current_stream(stream.device).cuda_stream)
            stream.wait_stream(cap)
            self._graphs[gkey] = graph
            entry = graph
        with torch.cuda.stream(stream):
            entry.replay()


class _Prepared:
    """Token of MonoRecModel.prepare(): the parsed inputs of one forward and its projection matrices."""

    def __init__(self, data, keyframe, frames, cv_depths, shape, device, kinv, proj, mats=None):
        self.data, self.keyframe, self.frames, self.cv_depths = data, keyframe, frames, cv_depths
        self.shape, self.device, self.kinv, self.proj, self.mats = shape, device, kinv, proj, mats


class _Group:
    """Requests collected by MonoRecModel.submit for one coalesced launch (hip_batch_keyframes > 1)."""

    def __init__(self, sig):
        self.sig, self.members, self.pending, self.error = sig, [], None, None


class _GroupHandle:
    """Handle of one member of a coalesced launch; same interface as _Pending."""

    def __init__(self, model, group, index):
        self._model, self._group, self._index = model, group, index

    def _launched(self):
        g = self._group
        if g.pending is None and g.error is None:
            self._model._flush_open_group()       # asked for before the group filled up: launch what has been collected
        if g.error is not None:                   # the combined launch failed (here or in the call that triggered it)
            raise RuntimeError(f"the coalesced launch of this request's group failed: {g.error!r}") from g.error
        return g.pending

    def result(self):
        self._launched().result()
        return self._group.members[self._index]

    def synchronize(self):
        self._launched().synchronize()
        return self._group.members[self._index]


class _Pending:
    """Handle of an enqueued forward (MonoRecModel.submit)."""

    def __init__(self, data_dict, done, device, consumers=None, owned=False, stream=None):
        self._data, self._done, self._device, self._consumers = data_dict, done, device, consumers
        self._stream = stream       # forward() on the caller's stream: results are already ordered on that stream
        self.collected = False      # result() / synchronize() taken: forward() may reuse the slot's resident output buffers as a copy source
        self.owned = owned          # the outputs already live in memory the caller owns (forward())

    def result(self):
        """Order the caller's current stream after the forward and return the output dict.  The stream is remembered: the submit
        that reuses the slot orders its launches behind whatever that stream has been given to do with the outputs by then."""
        cs = torch.cuda.current_stream(self._device)
        if self._stream is None or cs != self._stream:
            cs.wait_event(self._done)
        if self._consumers is not None and cs not in self._consumers:
            self._consumers.append(cs)
        self.collected = True
        return self._data

    def synchronize(self):
        """Wait on the HOST for the forward and return the output dict: the caller's stream needs no wait packet then (a blocked
        one slows the other hardware queues down) - the way to collect results in a pipelined loop."""
        _host_wait(self._done)
        self.collected = True
        return self._data


def _filter_state_dict(state_dict, data_parallel=False):
    """Key clean-up applied to reference checkpoints (utils/util.py:244-248): strip the DataParallel
    'module.' prefix, drop entries of list-style models 1-9 and strip a leading '0.'."""
    if data_parallel:
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    out = {}
    for k, v in state_dict.items():
        if k[0] in "123456789":
            continue
        out[k[2:] if k.startswith("0") else k] = v
    return out
