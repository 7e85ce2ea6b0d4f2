"""Launch plan for one (batch, H, W, frames, depth-steps) shape of the MonoRec inference path.

The plan owns every activation buffer (allocated once, HBM resident), the repacked weights and an
ordered list of C-ABI launches (include/monorec_hip.h).  It is the host-side replacement of the ATen
call sequence inside `MonoRecModel.forward` (reference model/monorec/monorec_model.py:672-729):

    stage "encoder" : ResnetEncoder.forward (:118-129) up to layer3    - independent of the poses
    stage "encoder_tail" : ResNet layer4 (output image_features[4] only; nothing downstream reads it)
    stage "cv"      : CostVolumeModule (:193-271) -> MaskModule encoder (:357-365)  - independent of the image features
    stage "main"    : MaskModule decoder (:370-383) -> (1-mask)*cv (:713) -> DepthModule (:526-557) -> affine (:717)
"encoder" and "cv" have no data dependence on each other, so MonoRecModel runs them on two HIP streams at the
same time (at batch 1 neither fills the 256 CUs alone); "main" joins them.

Because nothing in a stage allocates or synchronises, each stage can be recorded into a hipGraph
(torch.cuda.CUDAGraph is the HIP graph API on ROCm) and replayed with ~10 us of host cost instead of
~100 Python-driven launches.  The 4x4 pose algebra (:171,198,207) runs on the host between the two
stages, overlapped with the encoder stage (see MonoRecModel._forward_hip).
"""
import ctypes
import math
import re

import numpy as np
import torch

from . import _lib
from ._lib import (ACT_ABS_TANH_AFFINE, ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, IN_DIRECT, IN_MAXPOOL2,
                   IN_UPSAMPLE2, LAYOUT_BF16_B8, LAYOUT_F32_NCHW, TF_NONE, TF_RESNET_NORM, B8ConvDesc, ConvDesc, HeadDesc, WinoDesc)

LEAKY_SLOPE = 0.1   # model/layers.py:290,318,381
BN_EPS = 1e-5       # torch.nn.BatchNorm2d default, used by torchvision's ResNet


def same_pad(n, k, s):
    """TF 'same' padding of PadSameConv2d (model/layers.py:249-251): (low, high), floor on the low side."""
    total = s * (math.ceil(n / s) - 1) + k - n
    return math.floor(total / 2), math.ceil(total / 2)


def pack_conv_weight(weight, src_channels, mb, ck, bf16=False):
    """(Cout, sum(src_channels), kh, kw) fp32 CPU tensor -> packed A-fragment stream (CPU tensor) for a
    launch with `mb` cout blocks per workgroup and `ck` channels per LDS chunk (csrc/conv_layout.h);
    bf16 = compute mode: 0 / False fp32, 1 / True the bf16 stream of MR_COMPUTE_BF16 launches, 2 the hi / lo bf16 pairs of
    MR_COMPUTE_BF16X3."""
    lib = _lib.load()
    w = weight.detach().to(torch.float32).contiguous().cpu()
    cout, cin, kh, kw = w.shape
    assert cin == sum(src_channels), (cin, src_channels)
    sc = (ctypes.c_int32 * len(src_channels))(*src_channels)
    count, pack = {0: (lib.mr_conv_packed_weight_floats, lib.mr_conv_pack_weights_f32),
                   1: (lib.mr_conv_packed_weight_floats_bf16, lib.mr_conv_pack_weights_bf16),
                   2: (lib.mr_conv_packed_weight_floats_bf16x3, lib.mr_conv_pack_weights_bf16x3)}[int(bf16)]
    n = count(cout, sc, len(src_channels), kh, kw, mb, ck)
    assert n > 0, (mb, ck)
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(pack(w.data_ptr(), cout, sc, len(src_channels), kh, kw, mb, ck, out.data_ptr()), "mr_conv_pack_weights")
    return out


def fold_batchnorm(weight, sd, bn_prefix):
    """Eval-mode BatchNorm2d folded into the preceding bias-free conv: returns (weight', bias')."""
    g = sd[bn_prefix + ".weight"].double()
    beta = sd[bn_prefix + ".bias"].double()
    mean = sd[bn_prefix + ".running_mean"].double()
    var = sd[bn_prefix + ".running_var"].double()
    scale = g / torch.sqrt(var + BN_EPS)
    w = (weight.double() * scale.view(-1, 1, 1, 1)).float()
    b = (beta - mean * scale).float()
    return w, b


def transposed_phase_weights(wt):
    """ConvTranspose2d(k=4, s=2) + centre crop of 1 (layers.Refine, model/layers.py:389-397) as four
    2x2 stride-1 convolutions, one per output parity (py, px).

    wt: (Cin, Cout, 4, 4).  Output row y = 2*yy + py reads input rows yy - pad_top + t (t = 0, 1):
      py = 0: pad_top 1, taps ky = [3, 1];   py = 1: pad_top 0, taps ky = [2, 0]   (same for x).
    Returns {(py, px): (weight (Cout, Cin, 2, 2), pad_top, pad_left)}."""
    taps = {0: ([3, 1], 1), 1: ([2, 0], 0)}
    out = {}
    for py in (0, 1):
        for px in (0, 1):
            ky, pt = taps[py]
            kx, pl = taps[px]
            w = wt[:, :, ky, :][:, :, :, kx]            # (Cin, Cout, 2, 2)
            out[(py, px)] = (w.permute(1, 0, 2, 3).contiguous(), pt, pl)
    return out


def upconv_phase_weights(w):
    """layers.Upconv (model/layers.py:349-356): nearest x2 -> pad (0, 1, 0, 1) -> conv 2x2, as four convolutions on the
    LOW-resolution input, one per output parity (py, px).

    Output (2y + py, 2x + px) reads the upsampled rows 2y + py + dy, dy = 0, 1: for py = 0 both are input row y (the two filter
    rows add up), for py = 1 they are input rows y and y + 1 (zero below the image: the bottom pad); same for columns.  Parity
    (py, px) therefore is a (1 + py) x (1 + px) filter with pad 0 - 9 multiply-adds per 2x2 output block instead of 16.  The
    summed taps are formed in fp64 and rounded once (the reference adds the products instead; the difference is rounding
    noise of the fp32 accumulation, far inside the 1e-4 bar).  w: (Cout, Cin, 2, 2) -> {(py, px): (Cout, Cin, 1 + py, 1 + px)}."""
    wd = w.detach().double()
    out = {}
    for py in (0, 1):
        r = wd if py == 1 else wd[:, :, 0:1, :] + wd[:, :, 1:2, :]
        for px in (0, 1):
            c = r if px == 1 else r[:, :, :, 0:1] + r[:, :, :, 1:2]
            out[(py, px)] = c.float().contiguous()
    return out


def conv_geometry(out_h, out_w, kh, kw, sh, sw, nb, waves=4, kws=0):
    """Tile geometry used by mr_conv2d_f32 for a given NB / waves per workgroup (mirrors derive() in csrc/conv_mfma.hip).
    kws: the waves split K instead of the pixels - the workgroup owns nb pixel blocks instead of waves * nb."""
    blocks = nb if kws else waves * nb
    twb = 2 if (out_w >= 32 and blocks >= 2) else 1
    th = blocks // twb
    ih, iw = (th - 1) * sh + kh, (twb * 16 - 1) * sw + kw
    iw = (iw + 3 + 3) // 4 * 4          # upper bound: 4-aligned superset used by the dwordx4 DMA path
    plane = ih * iw
    if sw == 1:
        while plane % 32 != 16:
            plane += 1
    else:
        plane |= 1
    tiles = math.ceil(out_w / (twb * 16)) * math.ceil(out_h / th)
    return dict(twb=twb, th=th, ih=ih, iw=iw, plane=plane, tiles=tiles,
                tile_eff=(out_h * out_w) / (tiles * th * twb * 16), ppt=math.ceil(ih * iw / 256))


def lds_bytes(geo, taps, cpads, mb, ck, split_k=1, bf16=False, reduce_bytes=0):
    """Dynamic LDS of one workgroup: pipeline buffers of (input tile + A fragments of the largest chunk) - two,
    or one when no workgroup streams a second chunk (mirrors derive() in csrc/conv_mfma.hip); `reduce_bytes`: the scratch of the
    K-split-across-waves reduction (waves * mb * nb KiB), which reuses the same memory."""
    ck_max = max(min(c, ck) for c in cpads)
    nchunks = sum(math.ceil(c / ck) for c in cpads)
    nbuf = 2 if math.ceil(nchunks / split_k) > 1 else 1
    return max(nbuf * 4 * (ck * geo["plane"] + taps * ck_max * mb * (8 if int(bf16) == 1 else 16)), reduce_bytes)    # bf16x3 blocks: hi + lo = fp32 size


# Buffers of a plan that the model hands out as outputs (monorec_model.py:256-279,690,713-727).  `Plan.rebind_outputs` can point every
# launch that writes or reads them at caller-owned memory instead (MonoRecModel.forward: the outputs are produced where the caller
# keeps them - no copy), and back at the resident buffers (submit()).
OUTPUT_BUFFERS = ("cost_volume", "sfcv", "feat0", "feat1", "feat2", "feat3", "feat4", "cv_mask", "pred0", "pred1", "pred2", "pred3")


class _Ref:
    """Run-time pointer of a tensor captured by a launch closure: fixed, or `offset` bytes into an output buffer's CURRENT binding."""
    __slots__ = ("plan", "name", "offset", "fixed")

    def __init__(self, plan, name, offset, fixed):
        self.plan, self.name, self.offset, self.fixed = plan, name, offset, fixed

    def ptr(self):
        return self.fixed if self.name is None else self.plan.bound[self.name] + self.offset


# Environment variables that change WHICH kernels / forms a plan runs (and therefore its numerics and speed).  They exist for the tuning
# sessions (A/B of a candidate table, of one kernel family); a stray one must not pass silently: `active_env_overrides()` is
# echoed into bench.py's config block, and the first plan built under any of them warns once.
ENV_OVERRIDES = ("MR_TUNED_SCHEDULES", "MR_TUNED_WINOGRAD", "MR_TUNED_B8", "MR_B8", "MR_B8_NB4", "MR_B8_FEATS", "MR_WINOGRAD", "MR_ONE_CHANNEL_KERNELS", "MR_HIP_LIBRARY",
                 "MR_DIAG_STREAM_LAYOUT", "MR_DIAG_STREAM_PRIO")
_warned_env = [False]


def active_env_overrides():
    """{name: value} of the kernel-selection overrides set in this process's environment (empty = the committed tables and defaults)."""
    import os
    return {k: os.environ[k] for k in ENV_OVERRIDES if os.environ.get(k) not in (None, "")}


def _warn_env_overrides():
    if _warned_env[0]:
        return
    _warned_env[0] = True
    act = active_env_overrides()
    if act:
        import warnings
        warnings.warn(f"monorec_amd: kernel-selection overrides active in the environment: {act} - the plan differs from the committed "
                      "tables / defaults (tuning aid; unset them for the measured configuration)")


TUNED = {}          # signature -> (mb, nb, split_k, ck[, waves[, k_split_waves]]); filled from tuned_schedules.json when present


def _load_tuned():
    import json
    import os
    # (MR_TUNED_SCHEDULES / MR_TUNED_WINOGRAD: another table file - the tuning sessions A/B a candidate table before it replaces this one)
    path = os.environ.get("MR_TUNED_SCHEDULES") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_schedules.json")
    if os.path.exists(path):
        with open(path) as f:
            TUNED.update({k: tuple(v) for k, v in json.load(f).items()})


_load_tuned()

WINOGRAD = {}       # signature (without the phase / mode suffixes) -> 0 (direct kernel), 1 or 2 (cout blocks per wave; + 10: the variant
                    # with the input transform in registers); measured


def _load_winograd():
    import json
    import os
    path = os.environ.get("MR_TUNED_WINOGRAD") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_winograd.json")
    if os.path.exists(path):
        with open(path) as f:
            WINOGRAD.update({k: int(v) for k, v in json.load(f).items()})


_load_winograd()

B8_SCHEDULES = {}   # signature of a mr_conv2d_b8 launch (Plan.conv_b8: b8_co.._ci.._k.._s.._o.._b.._p.. + _f<fp32 source>) -> (MB, NB, waves); measured
                    # (tools/tune_b8.py) - launches without an entry follow the rule of Plan.b8_schedule


def _load_b8_schedules():
    import json
    import os
    path = os.environ.get("MR_TUNED_B8") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_b8.json")
    if os.path.exists(path):
        with open(path) as f:
            B8_SCHEDULES.update({k: tuple(int(x) for x in v) for k, v in json.load(f).items()})


_load_b8_schedules()

WINOGRAD_F2 = {}    # conv_forms = "f2": for a key whose table entry is a larger form (F(4,.), F(4x4,3x3)), the F(2,.) code that was
                    # measured best for it before the larger forms existed (0 = direct kernel: the 7-tap layers).  MEASURED for the c2 / c3
                    # keys only (15 of the 68 larger-form keys); every other key falls back to a rule - variant 11 for 3x3, the direct kernel
                    # for 1-D layers - whose throughput is unmeasured


def _load_winograd_f2():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_winograd_f2.json")
    if os.path.exists(path):
        with open(path) as f:
            WINOGRAD_F2.update({k: int(v) for k, v in json.load(f).items()})


_load_winograd_f2()


def winograd_signature(cout, src_channels, h, w, batch):
    return f"co{cout}_ci{'+'.join(str(c) for c in src_channels)}_o{h}x{w}_b{batch}"


_WSIG_RE = re.compile(r"^((?:[a-z]+\d*_)|(?:s2k\d+_))?co(\d+)_ci([\d+]+)_o(\d+)x(\d+)_b(\d+)$")
_parsed_wino = [None, 0]


def _wino_by_prefix():
    if _parsed_wino[0] is None or _parsed_wino[1] != len(WINOGRAD):
        by = {}
        for key, code in WINOGRAD.items():
            m = _WSIG_RE.match(key)
            if not m:
                continue
            cis = [int(c) for c in m.group(3).split("+")]
            by.setdefault(m.group(1) or "", []).append((int(m.group(2)), sum(cis), int(m.group(4)) * int(m.group(5)), int(m.group(6)), key, int(code)))
        _parsed_wino[0], _parsed_wino[1] = by, len(WINOGRAD)
    return _parsed_wino[0]


def nearest_form(prefix, cout, cin, pixels, batch, valid=lambda code: True, exclude=(), max_distance=4.0):
    """The kernel-form code (the WINOGRAD table's value) measured for the nearest signature with the same key prefix ('' 3x3, 'x_' / 'y_' / 'x7_' / 'y7_'
    1-D, 't_' Refine, 'u_' Upconv, 's2k<taps>_' stride-2 pairs) whose code is launchable for this layer (`valid`), or None: what a layer WITHOUT a table
    entry runs instead of the old "direct kernel unless the workgroup count says otherwise" rule.  Distance as in nearest_schedules."""
    best = None
    for (co, ci, px, b, key, code) in _wino_by_prefix().get(prefix, ()):
        if key in exclude or not valid(code):
            continue
        dist = abs(math.log2(co / cout)) + abs(math.log2(ci / cin)) + abs(math.log2(px / pixels)) + 0.5 * abs(math.log2(b / batch))
        if dist <= max_distance and (best is None or (dist, key) < best[:2]):
            best = (dist, key, code)
    return None if best is None else best[2]


def choose_winograd(cout, src_channels, h, w, batch, f2=False):
    """0 = direct MFMA kernel, 1 / 2 = Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) with 32 / 64 output channels per workgroup,
    11 / 12 = the same with the input transform in registers (mr_wino_desc.variant = 1), 21 = variant 2 (11 whose 1..16 tail channels
    come from 16-row workgroups: the 48-channel layers), 31 = F(4x4,3x3) (csrc/conv_wino44.hip; only by the table), 41 = F(4x4,3x3) with the
    positions of a tile split over two waves (csrc/conv_wino44s.hip: two workgroups per CU; only by the table), for a 3x3 stride-1 convolution.  The
    measured table (tools/bench_wino.py --emit, MI355X) wins; shapes it does not know go to the Winograd kernel when it has enough
    workgroups (8 x 32 output pixels each) to fill the chip - below that the direct kernel's smaller tiles and split-K win
    (measured: every ResNet layer of a batch-1 keyframe) - with the variant that measured faster at that width on every shape of
    the table (32 channels per workgroup: transform in registers, two workgroups per CU; 64: the LDS buffer)."""
    if w % 4:
        return 0
    sig = winograd_signature(cout, src_channels, h, w, batch)
    if sig in WINOGRAD:
        code = WINOGRAD[sig]
        if f2 and code // 10 in (3, 4, 5):      # F(4x4,3x3), either kernel -> the F(2x2,3x3) variant measured before it (else: transform in registers, 32 channels)
            code = WINOGRAD_F2.get(sig, 11)
        return code
    def ok(code):
        variant, mbw = code // 10, code % 10
        if code == 0:
            return True
        if variant == 2:
            return mbw == 1 and 0 < cout % 32 <= 16
        if variant in (0, 1):
            return mbw == 1 or cout > 32
        return variant in (3, 4, 5)
    code = nearest_form("", cout, sum(src_channels), h * w, batch, valid=ok)
    if code is not None:
        if f2 and code // 10 in (3, 4, 5):
            code = 11
        return code
    tiles = math.ceil(h / 8) * math.ceil(w / 32) * batch
    if tiles * math.ceil(cout / 32) < 256:
        return 0
    return 2 if (cout > 32 and tiles * math.ceil(cout / 64) >= 256) else 11



def choose_winograd_t(cout, src_channels, h, w, batch):
    """layers.Refine (ConvTranspose2d(4, 2) + crop) on an input of h x w: 0 = the four parity phases on the direct MFMA kernel, 1 / 2 /
    4 = the F(2x2,2x2) kernel (csrc/convt_wino.hip) with 32 / 64 / 128 output channels per workgroup, + 10 = its variant with the input
    transform in registers.  Only what the measured table (tools/bench_wino_t.py --emit; keys prefixed `t_`) says: shapes it does not
    know stay on the direct kernel."""
    if w % 4:
        return 0
    key = "t_" + winograd_signature(cout, src_channels, h, w, batch)
    if key in WINOGRAD:
        return WINOGRAD[key]
    return nearest_form("t_", cout, sum(src_channels), h * w, batch, valid=lambda c: c == 0 or (c % 10) * 32 <= max(32, cout)) or 0


def choose_winograd_1d(axis, cout, src_channels, h, w, batch, taps=3, f2=False):
    """k-tap stride-1 'same' convolution along x (axis 0: 1 x k) or y (axis 1: k x 1) - layers.ConvReLU2 of the DepthModule (3 taps: the
    second pair of every stage; 7 taps: enc.0.0): 0 = direct MFMA kernel; 1..4 = the 1-D Winograd F(2,3) kernel (csrc/conv1d_wino.hip)
    with 16 x that many output channels per workgroup; 10 m + mbw = the Cook-Toom form F(m, taps) of the same file (41..44: F(4,3);
    21..24 / 41..43 on the 7-tap keys: F(2,7) / F(4,7)).  Only what the measured table says (tools/bench_wino1d.py --emit; keys prefixed
    `x_` / `y_` for 3 taps, `x7_` / `y7_` for 7)."""
    if w % 4:
        return 0
    prefix = ("x", "y")[axis] + ("" if taps == 3 else str(taps)) + "_"
    key = prefix + winograd_signature(cout, src_channels, h, w, batch)
    if key in WINOGRAD:
        code = WINOGRAD[key]
    else:                                    # no entry: the form of the nearest measured signature of the same axis / tap count
        code = nearest_form(prefix, cout, sum(src_channels), h * w, batch, valid=lambda c: c == 0 or (c % 10) <= math.ceil(cout / 16)) or 0
        if f2 and code >= 40:
            return 0
    if f2 and code >= 40:                    # F(4,.) -> what was measured best among F(2,.) and the direct kernel before the larger forms
        code = WINOGRAD_F2.get(key, 0)
    return code


def stride2_signature(taps, cout, cin, out_h, out_w, batch):
    return f"s2k{taps}_co{cout}_ci{cin}_o{out_h}x{out_w}_b{batch}"


def choose_stride2(taps, cout, cin, out_h, out_w, batch):
    """A ConvReLU2 pair with stride 2 (k x 1 stride (2,1), then 1 x k stride (1,2): DepthModule.enc stages 1-3, monorec_model.py:489-501) on
    the stride-1 Cook-Toom kernel over [even | odd] views of its input (cooktoom.stride2_as_stride1; 7 taps: F(4,4), 5 taps: F(4,3)):
    10 * (blocks of 16 output channels per workgroup of the k x 1 half) + (the same of the 1 x k half; 0 = that half stays on the direct MFMA
    kernel), or 0 = both halves on the direct MFMA kernel.  Only what the measured table says (tools/bench_stride2.py --emit; keys `s2k<taps>_co<cout>_ci<cin>_o<out_h>x<out_w>_b<batch>`)."""
    if taps not in (5, 7) or out_w % 4:
        return 0
    key = stride2_signature(taps, cout, cin, out_h, out_w, batch)
    if key in WINOGRAD:
        return WINOGRAD[key]
    cb = math.ceil(cout / 16)
    return nearest_form(f"s2k{taps}_", cout, cin, out_h * out_w, batch, valid=lambda c: c == 0 or (c // 10 <= cb and c % 10 <= cb)) or 0


def stride2_unified_weights(w, n, axis):
    """(Cout, C, k, 1) [axis 1] / (Cout, C, 1, k) [axis 0] stride-2 filter -> the stride-1 filter over [even samples | odd samples]:
    (Cout, 2 C, r2, 1) / (Cout, 2 C, 1, r2) with r2 = ceil(k / 2) (cooktoom.stride2_as_stride1; a phase's missing tap is a zero weight)."""
    from . import cooktoom
    k = int(w.shape[2] if axis == 1 else w.shape[3])
    r2, pad, ev, od = cooktoom.stride2_as_stride1(k, n)
    assert pad == (r2 - 1) // 2, (k, pad)                          # the low-side padding the Cook-Toom kernel's geometry assumes ((R - 1) / 2)
    cout, c = int(w.shape[0]), int(w.shape[1])
    taps = w.reshape(cout, c, k)
    u = torch.zeros(cout, 2 * c, r2, dtype=torch.float32)
    for t in range(r2):
        if ev[t] is not None:
            u[:, :c, t] = taps[:, :, ev[t]]
        if od[t] is not None:
            u[:, c:, t] = taps[:, :, od[t]]
    return u.reshape(cout, 2 * c, r2, 1) if axis == 1 else u.reshape(cout, 2 * c, 1, r2)


def schedule_signature(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, bf16=False, mixed_phases=False):
    """`mixed_phases`: the phases sweep different tap counts (phase-decomposed Upconv: 1, 2, 2 and 4 taps of a 2x2 tile)."""
    return (f"co{cout}_ci{'+'.join(str(c) for c in src_channels)}_k{kh}x{kw}_s{sh}x{sw}_o{out_h}x{out_w}_b{batch}_p{phases}"
            + ("u" if mixed_phases else "") + ("", "_bf16", "_bf16x3")[int(bf16)])


def candidate_schedules(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases=1, lds_cap=80 * 1024, bf16=False):
    """All launchable (mb, nb, split_k, ck, waves, kws) for a conv, with the workgroup count of each."""
    cb = (cout + 15) // 16
    unit = 16 if bf16 else 4
    cpads = [(c + unit - 1) // unit * unit for c in src_channels]
    taps = kh * kw
    out = []
    for kws in (0, 1):
        if kws and bf16:
            continue                          # K split across the waves: fp32 launches only
        for waves in (4, 8):
            for nb in (4, 2, 1):
                geo = conv_geometry(out_h, out_w, kh, kw, sh, sw, nb, waves, kws)
                if waves == 4 and geo["ppt"] > 6:
                    continue
                if waves == 8 and geo["ih"] * (geo["iw"] // 4) > 256:      # dwordx4 groups per lane <= 4
                    continue
                for mb in (6, 4, 3, 2, 1):
                    if mb > cb and mb != 1:
                        continue
                    groups = math.ceil(cb / mb)
                    for ck in ((16, 32, 64, 128) if bf16 else (8, 16, 32, 64, 128)):
                        if ck > 16 and ck // 2 >= max(cpads):
                            continue
                        nchunks = sum(math.ceil(c / ck) for c in cpads)
                        wgs = geo["tiles"] * groups * batch * phases
                        if kws and wgs > 4096:
                            continue              # plenty of workgroups already: splitting K buys nothing
                        for sk in ((1,) if kws else (1, 2, 4, 8, 16)):
                            if sk > nchunks:
                                break
                            lds = lds_bytes(geo, taps, cpads, mb, ck, sk, bf16, waves * mb * nb * 1024 if kws else 0)
                            if lds > lds_cap:
                                continue
                            out.append(dict(mb=mb, nb=nb, split_k=sk, ck=ck, waves=waves, kws=kws, wgs=wgs * sk, nchunks=nchunks,
                                            eff=geo["tile_eff"] * cb / (groups * mb), lds=lds))
    return out


_SIG_RE = re.compile(r"^co(\d+)_ci([\d+]+)_k(\d+)x(\d+)_s(\d+)x(\d+)_o(\d+)x(\d+)_b(\d+)_p(\d+)(u?)(_bf16x3|_bf16)?$")
_parsed_tuned = [None, 0]


def _tuned_by_class():
    """TUNED parsed once: (kh, kw, sh, sw, phases, mixed, mode) -> [(cout, cin, nsrc, pixels, batch, key, schedule)]."""
    if _parsed_tuned[0] is None or _parsed_tuned[1] != len(TUNED):
        by = {}
        for key, sched in TUNED.items():
            m = _SIG_RE.match(key)
            if not m:
                continue
            cis = [int(c) for c in m.group(2).split("+")]
            cls = (int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(10)), m.group(11), m.group(12) or "")
            by.setdefault(cls, []).append((int(m.group(1)), sum(cis), len(cis), int(m.group(7)) * int(m.group(8)), int(m.group(9)), key, tuple(sched)))
        _parsed_tuned[0], _parsed_tuned[1] = by, len(TUNED)
    return _parsed_tuned[0]


def nearest_schedules(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases=1, bf16=False, mixed_phases=False, exclude=(), limit=6, max_distance=4.0):
    """Round 6 (the rule path was 29 % behind the tables on the one shape where both were measured): for a launch WITHOUT a table entry, the
    schedules measured for the nearest signatures of the same filter / stride / phase class, nearest first.  Distance = sum of |log2| ratios of output
    channels, input channels, output pixels and (half weight) batch, + 0.5 when the number of concatenated sources differs.  The caller validates each
    (LDS budget, chunk counts, DMA constraints are checked by the library) and falls back to the model."""
    cls = (kh, kw, sh, sw, phases, "u" if mixed_phases else "", ("", "_bf16", "_bf16x3")[int(bf16)])
    cin, nsrc, pix = sum(src_channels), len(src_channels), out_h * out_w
    scored = []
    for (co, ci, ns, px, b, key, sched) in _tuned_by_class().get(cls, ()):
        if key in exclude:
            continue
        dist = (abs(math.log2(co / cout)) + abs(math.log2(ci / cin)) + abs(math.log2(px / pix)) + 0.5 * abs(math.log2(b / batch)) + (0.5 if ns != nsrc else 0.0))
        if dist <= max_distance:
            scored.append((dist, key, sched))
    scored.sort()
    out, seen = [], set()
    for dist, key, sched in scored:
        if sched not in seen:
            seen.add(sched)
            out.append(sched)
        if len(out) >= limit:
            break
    return out


def schedule_candidates(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases=1, bf16=False, mixed_phases=False):
    """What Plan.conv tries, in order: the table entry of the signature alone when there is one; otherwise the nearest signatures' schedules, then the model."""
    sig = schedule_signature(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, bf16, mixed_phases)
    if sig in TUNED:
        return [TUNED[sig]]
    cands = nearest_schedules(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, bf16, mixed_phases)
    try:
        cands.append(tuple(choose_schedule(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, bf16, mixed_phases)))
    except ValueError:
        pass
    if not cands:
        raise ValueError("no launchable schedule")
    return cands


def choose_schedule(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases=1, bf16=False, mixed_phases=False):
    """(MB, NB, split_k, CK) for mr_conv2d_f32.  A tuned table (tools/tune_conv.py, measured on MI355X)
    wins; otherwise a model: biggest register tile that still puts >= 3 workgroups on each of the 256 CUs,
    deepest chunk that keeps >= 3 workgroups' LDS on a CU, split-K only to fill the machine."""
    sig = schedule_signature(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, bf16, mixed_phases)
    if sig in TUNED:
        return TUNED[sig]
    if int(bf16) == 2:
        # bf16x3 has no measured table yet: start from the schedule measured for the plain bf16 launch of the same layer (same
        # input staging, weight blocks twice as large), then from the fp32 one, whichever still fits the LDS
        cpads = [(c + 15) // 16 * 16 for c in src_channels]
        nchunks = lambda ck: sum(math.ceil(c / ck) for c in cpads)
        for other in (1, 0):
            t = TUNED.get(schedule_signature(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, other))
            if t is None or t[3] < 16 or t[2] > nchunks(t[3]):
                continue
            waves = t[4] if len(t) > 4 else 4
            geo = conv_geometry(out_h, out_w, kh, kw, sh, sw, t[1], waves)
            if lds_bytes(geo, kh * kw, cpads, t[0], t[3], t[2], 2) <= 160 * 1024:
                return t
    best = None
    for lds_cap in (80 * 1024, 160 * 1024):      # prefer two workgroups per CU; a one-per-CU budget only if nothing else launches
        for c in candidate_schedules(cout, src_channels, kh, kw, sh, sw, out_h, out_w, batch, phases, lds_cap=lds_cap, bf16=bf16):
            if c["waves"] != 4 or c.get("kws"):   # 8-wave / K-split-wave workgroups only through the measured table
                continue
            reuse = (c["mb"] * c["nb"]) / (c["mb"] + c["nb"])          # MFMAs per LDS operand read
            fill = min(1.0, c["wgs"] / 768.0)
            per_wg_steps = c["nchunks"] / c["split_k"]
            score = c["eff"] * fill * (0.5 + 0.5 * min(reuse, 1.5) / 1.5)
            score *= 1.0 - 0.04 * math.log2(c["split_k"])              # workspace round trip + extra launch
            score *= 1.0 + 0.03 * math.log2(c["ck"] / 16)              # fewer barriers
            if per_wg_steps < 2:
                score *= 0.8
            cand = (score, c["mb"], c["nb"], c["split_k"], c["ck"])
            if best is None or cand > best:
                best = cand
        if best is not None:
            break
    if best is None:
        raise ValueError("no launchable schedule")
    return best[1:]


class Plan:
    """Buffers + launch list for one input shape. `state` is a CPU state dict with the reference keys."""

    def __init__(self, state, batch, height, width, num_frames, depth_steps, inv_depth_min_max, device,
                 alpha=10.0, channel_weights=(5 / 32, 16 / 32, 11 / 32), schedule_override=None, build=True, bf16=False, use_ssim=True, sfcv_mult_mask=True,
                 pretrain_mode=0, no_cv=False, mask_use_cv=True, mask_use_feats=True, simple_mask=False, cv_patch_size=3,
                 one_channel_kernels=None, winograd=None, conv_forms="table", cv_separable=False, lean_outputs=False, skip_layer4=False):
        if build and (height % 32 or width % 32):
            raise ValueError("MonoRec needs height and width divisible by 32 (five stride-2 stages)")
        if build and depth_steps < 2:
            raise ValueError("cv_depth_steps must be >= 2 (monorec_model.py:258 divides by cv_depth_steps - 1)")
        self.lib = _lib.load()
        _warn_env_overrides()
        self.B, self.H, self.W, self.F, self.D = batch, height, width, num_frames, depth_steps
        self.device = torch.device(device)
        self.inv_depth_min_max = tuple(float(v) for v in inv_depth_min_max)
        self.alpha = float(alpha)
        self.cw = (ctypes.c_float * 3)(*[float(torch.tensor(c, dtype=torch.float32)) for c in channel_weights])
        self.schedule_override = schedule_override or {}
        self.cv_mode = {False: 0, True: 1}.get(use_ssim, use_ssim) if not isinstance(use_ssim, bool) else int(use_ssim)
        self.sfcv_mult_mask = bool(sfcv_mult_mask)
        self.pretrain_mode, self.no_cv = int(pretrain_mode), bool(no_cv)                  # monorec_model.py:680-727 (eval branches)
        self.mask_use_cv, self.mask_use_feats = bool(mask_use_cv), bool(mask_use_feats)   # :352-355
        self.simple_mask = bool(simple_mask)                                              # SimpleMaskModule, :388-473
        self.cv_patch_size = int(cv_patch_size)                                           # :138-142,247
        # opt-in (MonoRecModel(hip_cv_separable=True)): the default cost-volume configuration through mr_cost_volume_relaxed_f32 - separable
        # 3x3 window sums and x * fp32(1/9); validity identical, volumes within 1e-4, depth within 2e-6 of the exact-order kernel
        self.cv_separable = bool(cv_separable)
        # opt-in of the bf16 configuration (MonoRecModel(hip_bf16=True, hip_lean_outputs=True)): no dense fp32 `single_frame_cvs` - the fusion
        # kernel writes the fused volume and the B8 copies the mask encoder reads, nothing else (403 MB of HBM writes less at configs[4])
        self.lean_outputs = bool(lean_outputs)
        # opt-in (MonoRecModel(hip_skip_dead_layer4=True)): ResNet layer4 (monorec_model.py:118-129) is computed by the reference but read by nobody -
        # MaskModule / DepthModule consume image_features[0..3] only (:372-380, :545).  With the switch its 5 convolutions + 3 split-K
        # finishing launches are not issued and `image_features` has four entries.
        self.skip_layer4 = bool(skip_layer4)
        self.pix_depths_on = False    # set per forward by the model when the input dict carries per-pixel cv_depths
        self.bf16 = int(bf16)         # convolutions: 0 fp32 MFMA, 1 bf16 MFMA (MR_COMPUTE_BF16), 2 bf16x3 split (MR_COMPUTE_BF16X3); everything else fp32
        # bf16 MFMA mode: the activations BETWEEN the convolutions of the mask and depth nets are stored channel-blocked in bf16 ("B8",
        # csrc/conv_b8.hip) - half the HBM bytes, the MFMA operand read from LDS as it is.  MR_B8=0: A/B aid (fp32 storage as in rounds 1-3)
        import os as _os0
        self.b8 = self.bf16 == 1 and _os0.environ.get("MR_B8", "1") != "0"
        self.b8_feature_copies = _os0.environ.get("MR_B8_FEATS", "1") != "0"      # MR_B8_FEATS=0: A/B aid (decoders read the fp32 image features, rounds 4-5)
        self.sd = state
        self.buf = {}
        self.keep = []          # packed weights / biases (device tensors kept alive)
        self.stages = {"encoder": [], "encoder_tail": [], "cv": [], "main": []}
        self.conv_log = []      # (name, macs, mb, nb, split_k, wgs) for bench / tuning
        self.aux_log = []       # one-channel layers that run on their own HBM-bound kernels (csrc/heads.hip): name, ref_macs
        if one_channel_kernels is None:         # MR_ONE_CHANNEL_KERNELS=0: A/B aid - classifier / depth heads as mr_conv2d_f32 launches
            import os
            one_channel_kernels = os.environ.get("MR_ONE_CHANNEL_KERNELS", "1") != "0"
        self.one_channel_kernels = bool(one_channel_kernels)
        import os as _os
        # 3x3 stride-1 convolutions on the Winograd F(2x2,3x3) kernel where it is faster (choose_winograd); MR_WINOGRAD=0: A/B aid
        self.winograd = _os.environ.get("MR_WINOGRAD", "1") != "0" if winograd is None else bool(winograd)
        # which reduced-multiply forms the measured table may select (MonoRecModel(hip_exact_convs=...)):
        #   "table"  everything it holds, F(4,3) / F(4,7) / F(4x4,3x3) included (transform constants up to 89 and 1/2835);
        #   "f2"     the F(2,.) forms only (constants 0, +-1, +-1/2: rounding like the direct sum) - a layer the table gives to a larger
        #            form runs what was measured best for it before those forms existed (tuned_winograd_f2.json);
        #   "direct" none: every convolution on the direct MFMA kernel, an exact fmaf chain per output like the reference's.
        if conv_forms not in ("table", "f2", "direct"):
            raise ValueError(f"conv_forms must be 'table', 'f2' or 'direct', got {conv_forms!r}")
        self.conv_forms = conv_forms
        if conv_forms == "direct":
            self.winograd = False
        self.input_ptr = {}       # "keyframe" -> device pointer the launches read the keyframe from (resident copy or the caller's tensor)
        self._input_srcs = []     # (ConvDesc, source index, "keyframe"): descriptor slots that follow input_ptr
        self._frame_ptrs = None   # ctypes array of the F source-frame pointers handed to the cost-volume launch
        self._compiled = {}       # stage -> (launches compiled, [closure | (mr_launch_item array, count, names)]): see run_stage
        self._ws_floats = {}      # stage -> floats: stages may run concurrently on different streams,
        self._pending_ws = []     # so every stage gets its own split-K workspace
        self.bound = {}           # output buffer name -> base pointer its launches currently use (rebind_outputs)
        self._relocs = []         # (ctypes object, field name, index or None, output buffer name, byte offset): see _index_outputs
        self._ptr_arrays = []     # ctypes pointer arrays handed to launches (scanned by _index_outputs)
        if build:
            self._build()
            self.finalize()

    @classmethod
    def bare(cls, device, state=None, schedule_override=None, bf16=False):
        """Plan without the network: lets tests / micro-benchmarks launch single ops through the C ABI."""
        return cls(state or {}, 1, 32, 32, 1, 4, (0.33, 0.0025), device, schedule_override=schedule_override, build=False, bf16=bf16)

    def launch_stamp(self):
        """What THIS plan launches: sha256 over (layer, kernel family / variant, schedule) of every convolution launch + the
        library's ABI version.  tools/summarize_prof.py stamps a committed profile set with the stamp of the plan that was profiled
        (taken from the bench line of the profiled run); bench.py quotes a committed rocprof figure only when that stamp equals the
        stamp of the plan it is timing (the round-3 driver line quoted profiles one table behind HEAD).  Table entries
        for OTHER shapes do not move it."""
        import hashlib
        h = hashlib.sha256()
        for c in self.conv_log:
            h.update(repr((c["name"], c.get("winograd", 0), c.get("wino_variant", 0), c.get("wino_axis", 0), c.get("wino_m", 0), c.get("b8", 0),
                           c["mb"], c["nb"], c["split_k"], c["ck"], c["waves"], c["kws"], int(c.get("bf16", 0)))).encode())
        h.update(f"abi{_lib.MR_ABI_VERSION}".encode())
        return h.hexdigest()[:16]

    def finalize(self):
        """Allocate the per-stage split-K workspace once all launches are known."""
        for stage, floats in self._ws_floats.items():
            self.buf[f"splitk_workspace.{stage}"] = torch.empty(max(floats, 4), dtype=torch.float32, device=self.device)
        for stage, desc in self._pending_ws:
            desc.workspace = self.buf[f"splitk_workspace.{stage}"].data_ptr()
        self._pending_ws = []
        self._ws_floats = {}
        self._index_outputs()

    # ------------------------------------------------------------------ output binding
    def _out_range(self, ptr):
        """(name, byte offset) of the output buffer `ptr` points into, or (None, 0)."""
        for name in OUTPUT_BUFFERS:
            t = self.buf.get(name)
            if t is not None:
                lo = t.data_ptr()
                if lo <= ptr < lo + t.numel() * t.element_size():
                    return name, ptr - lo
        return None, 0

    def ref(self, t):
        """Pointer of `t` as a launch closure should read it at run time (follows rebind_outputs when `t` lies in an output buffer)."""
        name, off = self._out_range(t.data_ptr())
        return _Ref(self, name, off, t.data_ptr())

    def _index_outputs(self):
        """Relocation table of the output buffers: every pointer slot of a descriptor (or of a pointer array handed to a launch) that
        points into one of OUTPUT_BUFFERS, found by address.  Descriptors are referenced by the native launch lists, so patching
        them in place re-targets the launches."""
        self.bound = {n: self.buf[n].data_ptr() for n in OUTPUT_BUFFERS if n in self.buf}
        self._resident = dict(self.bound)
        self._relocs = []
        seen = set()

        def slot(obj, field, idx, ptr):
            if not ptr:
                return
            name, off = self._out_range(int(ptr))
            if name is not None:
                self._relocs.append((obj, field, idx, name, off))
        for obj in self.keep:
            if id(obj) in seen:
                continue
            seen.add(id(obj))
            if isinstance(obj, (ConvDesc, WinoDesc)):
                for i in range(obj.num_src):
                    slot(obj, "src", i, obj.src[i])
                slot(obj, "dst", None, obj.dst)
                slot(obj, "residual", None, obj.residual)
            elif isinstance(obj, B8ConvDesc):
                for i in range(obj.num_src):
                    slot(obj, "src", i, obj.src[i])
                slot(obj, "dst", None, obj.dst)
            elif isinstance(obj, ctypes.Array) and getattr(obj, "_type_", None) is HeadDesc:
                for i in range(len(obj)):
                    slot(obj[i], "src", None, obj[i].src)
                    slot(obj[i], "dst", None, obj[i].dst)
        for arr in self._ptr_arrays:
            for i in range(len(arr)):
                slot(arr, None, i, arr[i])

    def rebind_outputs(self, bases=None):
        """Point every launch at `bases[name]` (device pointers of caller-owned memory, one per output buffer, same layout) - or, with
        None, back at the plan's resident buffers.  Only what changes is patched."""
        target = self._resident if bases is None else {n: bases.get(n, self._resident[n]) for n in self._resident}
        if target == self.bound:
            return
        changed = {n for n in target if target[n] != self.bound[n]}
        for obj, field, idx, name, off in self._relocs:
            if name in changed:
                p = target[name] + off
                if field is None:
                    obj[idx] = p
                elif idx is None:
                    setattr(obj, field, p)
                else:
                    getattr(obj, field)[idx] = p
        self.bound = dict(target)

    @property
    def outputs_rebindable(self):
        """Plans whose output buffers are written by every forward (the full model): the variants that keep constant content in
        them (no_cv zero volumes, pretrain_mode 1 / 3 masks, mask-only mode) hand out copies of the resident buffers instead."""
        preds = getattr(self, "preds", None)
        return self.pretrain_mode == 0 and not self.no_cv and preds is not None and all(p is not None for p in preds)

    # ------------------------------------------------------------------ buffers / parameters
    def alloc(self, name, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        self.buf[name] = t
        return t

    def _dev(self, t):
        d = t.detach().to(torch.float32).contiguous().to(self.device)
        self.keep.append(d)
        return d

    # ------------------------------------------------------------------ op builders
    def _launch_conv(self, desc, name):
        lib = self.lib

        def run(stream):
            _lib.check(lib.mr_conv2d_f32(ctypes.byref(desc), stream), name)
        run.native = (_lib.LAUNCH_CONV2D, desc, 0)          # run_stage walks runs of such launches in one host call (mr_run_launches)
        return run

    def conv(self, stage, name, srcs, weight, bias, out, *, stride=(1, 1), pad=(0, 0), grid=None,
             act=ACT_NONE, p0=0.0, p1=0.0, in_mode=IN_DIRECT, tf=TF_NONE, residual=None,
             out_step=(1, 1), out_off=(0, 0), out_ch_offset=0, phases=None, ref_macs=None):
        """Append one mr_conv2d_f32 launch. srcs: list of (N,C,Hs,Ws) tensors concatenated on channels.
        phases: optional list of 4 (weight, pad_top, pad_left, out_off_h, out_off_w) run in one launch."""
        n, _, hs, ws = srcs[0].shape
        src_channels = [int(s.shape[1]) for s in srcs]
        for s in srcs:
            assert s.shape[0] == n and s.shape[2] == hs and s.shape[3] == ws and s.is_contiguous()
        w0 = weight if phases is None else phases[0][0]
        cout, cin, kh, kw = w0.shape
        if (self.winograd and phases is None and (kh, kw) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and
                in_mode == IN_DIRECT and tf == TF_NONE and tuple(out_step) == (1, 1) and tuple(out_off) == (0, 0) and out_ch_offset == 0 and
                out.shape[1] == cout and tuple(grid) == (hs, ws) and act in (ACT_NONE, ACT_RELU, ACT_LEAKY_RELU) and self.bf16 == 0 and
                name not in self.schedule_override):
            mbw = choose_winograd(cout, src_channels, hs, ws, n, f2=self.conv_forms == "f2")
            if mbw:
                return self._conv_winograd(stage, name, srcs, weight, bias, out, act, p0, residual, mbw % 10, mbw // 10)
        if (self.winograd and phases is None and (kh, kw) in ((1, 3), (3, 1), (1, 7), (7, 1)) and tuple(stride) == (1, 1) and tuple(pad) == (kh // 2, kw // 2) and
                in_mode == IN_DIRECT and tf == TF_NONE and tuple(out_step) == (1, 1) and tuple(out_off) == (0, 0) and out_ch_offset == 0 and
                out.shape[1] == cout and tuple(grid) == (hs, ws) and act in (ACT_NONE, ACT_RELU, ACT_LEAKY_RELU) and self.bf16 == 0 and
                residual is None and name not in self.schedule_override):
            axis = 0 if kh == 1 else 1
            code = choose_winograd_1d(axis, cout, src_channels, hs, ws, n, max(kh, kw), f2=self.conv_forms == "f2")
            if code:
                return self._conv_winograd_1d(stage, name, srcs, weight, bias, out, act, p0, axis, code % 10, code // 10 if code >= 10 else 2)
        if phases is not None:                 # the common kh x kw sizes the input tile: the maximum over the phases
            kh, kw = max(p[0].shape[2] for p in phases), max(p[0].shape[3] for p in phases)
        mixed = phases is not None and any(tuple(p[0].shape[2:]) != (kh, kw) for p in phases)
        assert cin == sum(src_channels), (name, cin, src_channels)
        out_h, out_w = grid
        nph = 1 if phases is None else len(phases)
        bf16 = self.bf16 if (in_mode != IN_MAXPOOL2 and tf == TF_NONE) else 0   # the bf16 modes need the LDS-DMA staging
        cands = ([self.schedule_override[name]] if self.schedule_override.get(name) else
                 schedule_candidates(cout, src_channels, kh, kw, stride[0], stride[1], out_h, out_w, n, nph, bf16, mixed))
        d = ConvDesc()
        for i, s in enumerate(srcs):
            d.src[i] = s.data_ptr()
            d.src_channels[i] = src_channels[i]
            if "keyframe" in self.buf and s is self.buf["keyframe"]:
                self._input_srcs.append((d, i, "keyframe"))
        d.num_src, d.batch, d.src_h, d.src_w = len(srcs), n, hs, ws
        d.in_mode, d.in_transform = in_mode, tf
        d.kh, d.kw, d.stride_h, d.stride_w, d.pad_top, d.pad_left = kh, kw, stride[0], stride[1], pad[0], pad[1]
        d.out_h, d.out_w = out_h, out_w
        d.dst = out.data_ptr()
        assert out.shape[0] == n and out.is_contiguous()
        d.out_channels, d.dst_total_channels, d.dst_channel_offset = cout, out.shape[1], out_ch_offset
        d.dst_plane_h, d.dst_plane_w = out.shape[2], out.shape[3]
        d.out_step_h, d.out_step_w, d.out_off_h, d.out_off_w = out_step[0], out_step[1], out_off[0], out_off[1]
        d.num_phases = 1 if phases is None else nph
        if phases is not None:
            for i, (wp, pt, pl, oh, ow) in enumerate(phases):
                d.phase_pad_top[i], d.phase_pad_left[i], d.phase_out_off_h[i], d.phase_out_off_w[i] = pt, pl, oh, ow
                d.phase_kh[i], d.phase_kw[i] = wp.shape[2], wp.shape[3]
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        if residual is not None:
            assert residual.shape == out.shape
            d.residual = residual.data_ptr()
        d.activation, d.act_p0, d.act_p1 = act, p0, p1
        d.compute_dtype = int(bf16)
        # the first candidate the library itself accepts (LDS budget, chunk counts, DMA constraints: mr_conv2d_lds_bytes validates the whole descriptor);
        # a table entry or an override is the only candidate and must launch
        lds, sched = -1, None
        for sched in cands:
            mb, nb, split_k, ck = sched[:4]
            waves = sched[4] if len(sched) > 4 else 4
            kws = int(sched[5]) if len(sched) > 5 else 0       # K split across the waves of a workgroup
            d.cout_blocks_per_wg, d.pixel_blocks_per_wave, d.split_k, d.chunk_channels = mb, nb, split_k, ck
            d.waves_per_wg, d.k_split_waves = waves, kws
            d.workspace = 1 if split_k > 1 else None           # placeholder (non-null) until the shared workspace exists
            d.packed_weights = 1                               # (placeholders: only the geometry is validated here)
            for i in range(nph if phases is not None else 0):
                d.phase_weights[i] = 1
            lds = self.lib.mr_conv2d_lds_bytes(ctypes.byref(d))
            if lds >= 0:
                break
        if lds < 0:
            _lib.check(int(lds), f"plan {name} sched={sched}")
        if phases is None:
            d.packed_weights = self._dev(pack_conv_weight(weight, src_channels, mb, ck, bf16)).data_ptr()
        else:
            for i, (wp, pt, pl, oh, ow) in enumerate(phases):
                d.phase_weights[i] = self._dev(pack_conv_weight(wp, src_channels, mb, ck, bf16)).data_ptr()
        if split_k > 1:
            self._ws_floats[stage] = max(self._ws_floats.get(stage, 0),
                                         split_k * nph * n * ((cout + 15) // 16 * 16) * out_h * out_w)
            self._pending_ws.append((stage, d))
        taps = kh * kw if phases is None else sum(p[0].shape[2] * p[0].shape[3] for p in phases)
        macs = n * out_h * out_w * cout * cin * taps
        geo = conv_geometry(out_h, out_w, kh, kw, stride[0], stride[1], nb, waves, kws)
        wgs = geo["tiles"] * math.ceil(((cout + 15) // 16) / mb) * n * split_k * nph
        self.conv_log.append(dict(name=name, macs=macs, ref_macs=macs if ref_macs is None else ref_macs, mb=mb, nb=nb, split_k=split_k, ck=ck, waves=waves, kws=kws, wgs=wgs, lds=int(lds),
                                  cout=cout, cin=cin, k=(kh, kw), out=(out_h, out_w), batch=n, phases=nph,
                                  sig=schedule_signature(cout, src_channels, kh, kw, stride[0], stride[1], out_h, out_w, n, nph, bf16, mixed), bf16=bf16,
                                  spec=dict(src_shapes=[tuple(s.shape) for s in srcs], w_shape=(cout, cin, kh, kw),
                                            stride=tuple(stride), pad=tuple(pad), grid=(out_h, out_w), in_mode=in_mode,
                                            tf=tf, act=act, p0=p0, p1=p1, residual=residual is not None,
                                            out_shape=tuple(out.shape), out_step=tuple(out_step), out_off=tuple(out_off),
                                            phases=None if phases is None else [(pt, pl, oh, ow, int(wp.shape[2]), int(wp.shape[3]))
                                                                                for wp, pt, pl, oh, ow in phases])))
        self.keep += [d, out, residual] + list(srcs)      # the descriptor only holds raw pointers
        self.stages[stage].append((name, self._launch_conv(d, name)))
        return out

    def _conv_winograd(self, stage, name, srcs, weight, bias, out, act, p0, residual, mbw, variant=0):
        """One mr_conv3x3_winograd_f32 launch (csrc/conv_wino.hip) in place of a 3x3 stride-1 mr_conv2d_f32 launch."""
        lib = self.lib
        n, _, hs, ws = srcs[0].shape
        src_channels = [int(s.shape[1]) for s in srcs]
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        sc = (ctypes.c_int32 * len(src_channels))(*src_channels)
        w = weight.detach().to(torch.float32).contiguous().cpu()
        if variant == 4:       # F(4x4,3x3) with the positions split over two waves (csrc/conv_wino44s.hip): 8 x 64 pixels x 32 channels, two workgroups per CU
            nfl = lib.mr_wino44s_packed_weight_floats(cout, sc, len(src_channels))
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino44s_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), packed.data_ptr()), "mr_wino44s_pack_weights_f32")
        elif variant in (3, 5):     # F(4x4,3x3) (csrc/conv_wino44.hip; 5: csrc/conv_wino44w.hip, one wave per SIMD): 36 positions, 16 x 64 pixels x 32 channels per workgroup
            nfl = lib.mr_wino44_packed_weight_floats(cout, sc, len(src_channels))
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino44_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), packed.data_ptr()), "mr_wino44_pack_weights_f32")
        elif variant == 2:     # 32 a + (1..16) output channels: the tail group by 16-row workgroups (csrc/conv_wino.hip: wino_rb_tail)
            assert mbw == 1 and 0 < cout % 32 <= 16, (name, cout)
            nfl = lib.mr_wino_packed_weight_floats_tail(cout, sc, len(src_channels))
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino_pack_weights_tail_f32(w.data_ptr(), cout, sc, len(src_channels), packed.data_ptr()), "mr_wino_pack_weights_tail_f32")
        else:
            nfl = lib.mr_wino_packed_weight_floats(cout, sc, len(src_channels), mbw)
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), mbw, packed.data_ptr()), "mr_wino_pack_weights_f32")
        d = WinoDesc()
        for i, s_ in enumerate(srcs):
            d.src[i], d.src_channels[i] = s_.data_ptr(), src_channels[i]
            if "keyframe" in self.buf and s_ is self.buf["keyframe"]:
                self._input_srcs.append((d, i, "keyframe"))
        d.num_src, d.batch, d.height, d.width = len(srcs), n, hs, ws
        d.dst, d.out_channels = out.data_ptr(), cout
        assert out.is_contiguous() and tuple(out.shape) == (n, cout, hs, ws)
        d.packed_weights = self._dev(packed).data_ptr()
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        if residual is not None:
            assert residual.shape == out.shape
            d.residual = residual.data_ptr()
        d.activation, d.act_p0, d.cout_blocks_per_wave, d.variant = act, p0, mbw, variant
        lds = (lib.mr_conv3x3_winograd44s_lds_bytes(ctypes.byref(d)) if variant == 4 else lib.mr_conv3x3_winograd44w_lds_bytes(ctypes.byref(d)) if variant == 5 else
               lib.mr_conv3x3_winograd44_lds_bytes(ctypes.byref(d)) if variant == 3 else lib.mr_conv3x3_winograd_lds_bytes(ctypes.byref(d)))
        if lds < 0:
            _lib.check(int(lds), f"plan {name} winograd")
        ref = n * hs * ws * cout * cin * 9
        if variant == 4:
            wgs = math.ceil(hs / 8) * math.ceil(ws / 64) * n * math.ceil(cout / 32)
        elif variant in (3, 5):
            wgs = math.ceil(hs / 16) * math.ceil(ws / 64) * n * math.ceil(cout / 32)
        else:
            wgs = math.ceil(hs / 8) * math.ceil(ws / 32) * n * math.ceil(cout / (32 * mbw))
        self.conv_log.append(dict(name=name, macs=ref // 4 if variant in (3, 4, 5) else ref * 4 // 9, ref_macs=ref, mb=mbw, nb=0, split_k=1, ck=4 if variant in (4, 5) else 8, waves=4 if variant == 5 else 8, kws=0, wgs=wgs, lds=int(lds),
                                  cout=cout, cin=cin, k=(3, 3), out=(hs, ws), batch=n, phases=1, winograd=mbw, wino_variant=variant, bf16=0,
                                  sig=winograd_signature(cout, src_channels, hs, ws, n),
                                  spec=dict(src_shapes=[tuple(s_.shape) for s_ in srcs], w_shape=(cout, cin, 3, 3), stride=(1, 1), pad=(1, 1),
                                            grid=(hs, ws), in_mode=IN_DIRECT, tf=TF_NONE, act=act, p0=p0, p1=0.0, residual=residual is not None,
                                            out_shape=tuple(out.shape), out_step=(1, 1), out_off=(0, 0), phases=None)))
        self.keep += [d, out, residual] + list(srcs)

        if variant == 4:
            def run(stream):
                _lib.check(lib.mr_conv3x3_winograd44s_f32(ctypes.byref(d), stream), name)
            run.native = (_lib.LAUNCH_WINO44S, d, 0)
        elif variant == 5:
            def run(stream):
                _lib.check(lib.mr_conv3x3_winograd44w_f32(ctypes.byref(d), stream), name)
            run.native = (_lib.LAUNCH_WINO44W, d, 0)
        elif variant == 3:
            def run(stream):
                _lib.check(lib.mr_conv3x3_winograd44_f32(ctypes.byref(d), stream), name)
            run.native = (_lib.LAUNCH_WINO44, d, 0)
        else:
            def run(stream):
                _lib.check(lib.mr_conv3x3_winograd_f32(ctypes.byref(d), stream), name)
            run.native = (_lib.LAUNCH_WINO3X3, d, 0)
        self.stages[stage].append((name, run))
        return out

    def _conv_winograd_1d(self, stage, name, srcs, weight, bias, out, act, p0, axis, mbw, m=2, view=None, dst_split=False, ref_macs=None, ref_k=None,
                          sig=None):
        """One mr_conv1d3_winograd_f32 launch (csrc/conv1d_wino.hip: F(2,3), 4 instead of 6 multiplies per output pair) - or, for m = 4 or
        7 taps, one mr_conv1d_cooktoom_f32 launch (F(m, taps): m + taps - 1 multiplies per m outputs) - in place of a k x 1 / 1 x k
        stride-1 mr_conv2d_f32 launch."""
        lib = self.lib
        if view is None:
            n, _, hs, ws = srcs[0].shape
            src_channels = [int(s_.shape[1]) for s_ in srcs]
            src_ptrs = [s_.data_ptr() for s_ in srcs]
        else:
            # strided source views (the k x 1 stride-(2,1) half of a ConvReLU2 pair as a stride-1 form over [even rows | odd rows]):
            # view = dict(ptrs, channels, batch, height, width, row_pitch, plane) - `srcs` only keeps the underlying tensors alive
            n, hs, ws = view["batch"], view["height"], view["width"]
            src_channels, src_ptrs = list(view["channels"]), list(view["ptrs"])
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        taps = int(weight.shape[3] if axis == 0 else weight.shape[2])
        assert tuple(weight.shape[2:]) == ((1, taps) if axis == 0 else (taps, 1)) and cin == sum(src_channels), (name, weight.shape)
        general = (m, taps) != (2, 3)
        sc = (ctypes.c_int32 * len(src_channels))(*src_channels)
        w = weight.detach().to(torch.float32).contiguous().cpu()
        if general:
            nfl = lib.mr_cooktoom1d_packed_weight_floats(cout, sc, len(src_channels), mbw, m, taps)
            assert nfl > 0, (name, m, taps, mbw)
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_cooktoom1d_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), mbw, m, taps, packed.data_ptr()),
                       "mr_cooktoom1d_pack_weights_f32")
        else:
            nfl = lib.mr_wino1d_packed_weight_floats(cout, sc, len(src_channels), mbw)
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino1d_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), mbw, packed.data_ptr()), "mr_wino1d_pack_weights_f32")
        d = WinoDesc()
        for i in range(len(src_ptrs)):
            d.src[i], d.src_channels[i] = src_ptrs[i], src_channels[i]
            if view is None and "keyframe" in self.buf and srcs[i] is self.buf["keyframe"]:
                self._input_srcs.append((d, i, "keyframe"))
        d.num_src, d.batch, d.height, d.width = len(src_ptrs), n, hs, ws
        if view is not None:
            assert general, name
            d.src_row_pitch, d.src_plane_floats = int(view["row_pitch"]), int(view["plane"])
        if dst_split:            # (2, n, cout, hs, ws / 2): even columns, odd columns - the two sources of the 1 x k stride-(1,2) half
            assert general and axis == 1 and out.is_contiguous() and tuple(out.shape) == (2, n, cout, hs, ws // 2), (name, tuple(out.shape))
            d.dst_split_columns = 1
        else:
            assert out.is_contiguous() and tuple(out.shape) == (n, cout, hs, ws)
        d.dst, d.out_channels = out.data_ptr(), cout
        d.packed_weights = self._dev(packed).data_ptr()
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        d.activation, d.act_p0, d.cout_blocks_per_wave = act, p0, mbw
        lds = lib.mr_conv1d_cooktoom_lds_bytes(ctypes.byref(d), axis, m, taps) if general else lib.mr_conv1d3_winograd_lds_bytes(ctypes.byref(d))
        if lds < 0:
            _lib.check(int(lds), f"plan {name} winograd-1d")
        ref = n * hs * ws * cout * cin * taps
        rh, rw = ((8, 16 * m) if axis == 0 else (4 * m, 32)) if general else (8, 32)
        wgs = math.ceil(hs / rh) * math.ceil(ws / rw) * n * math.ceil(cout / (16 * mbw))
        kk = (1, taps) if axis == 0 else (taps, 1)
        # (ref_macs / ref_k: the reference's multiply-adds and filter of a stride-2 layer run as a stride-1 form over [even | odd] - the unified
        # filter has 2 * ceil(k / 2) >= k taps per input channel, of which the padded one is a zero weight)
        self.conv_log.append(dict(name=name, macs=ref * (m + taps - 1) // (m * taps), ref_macs=ref if ref_macs is None else ref_macs, mb=mbw, nb=0, split_k=1, ck=8,
                                  waves=8, kws=0, wgs=wgs,
                                  lds=int(lds), cout=cout, cin=cin, k=kk if ref_k is None else ref_k, out=(hs, ws), batch=n, phases=1, winograd=mbw, wino_variant=0,
                                  wino_axis=axis, wino_m=m, wino_taps=taps, stride2=ref_k is not None, bf16=0,
                                  sig=sig or (("x", "y")[axis] + ("" if taps == 3 else str(taps)) + "_" + winograd_signature(cout, src_channels, hs, ws, n)),
                                  spec=dict(src_shapes=[(n, c_, hs, ws) for c_ in src_channels], w_shape=(cout, cin) + kk,
                                            stride=(1, 1) if ref_k is None else ((2, 1) if axis == 1 else (1, 2)), pad=(kk[0] // 2, kk[1] // 2),
                                            grid=(hs, ws), in_mode=IN_DIRECT, tf=TF_NONE, act=act, p0=p0, p1=0.0, residual=False,
                                            out_shape=tuple(out.shape), out_step=(1, 1), out_off=(0, 0), phases=None)))
        self.keep += [d, out] + list(srcs)

        if general:
            def run(stream):
                _lib.check(lib.mr_conv1d_cooktoom_f32(ctypes.byref(d), axis, m, taps, stream), name)
            run.native = (_lib.LAUNCH_COOKTOOM_1D, d, axis | (m << 4) | (taps << 8))
        else:
            def run(stream):
                _lib.check(lib.mr_conv1d3_winograd_f32(ctypes.byref(d), axis, stream), name)
            run.native = (_lib.LAUNCH_WINO_1D, d, axis)
        self.stages[stage].append((name, run))
        return out

    def same_conv(self, stage, name, srcs, wkey, bkey, out, *, stride=(1, 1), act=ACT_LEAKY_RELU, p0=LEAKY_SLOPE,
                  p1=0.0, in_mode=IN_DIRECT):
        """PadSameConv2d + Conv2d (+ activation): model/layers.py:241-252,329-335."""
        w = self.sd[wkey]
        hs, ws = srcs[0].shape[2], srcs[0].shape[3]
        if in_mode == IN_UPSAMPLE2:
            hin, win = 2 * hs, 2 * ws
        elif in_mode == IN_MAXPOOL2:
            hin, win = hs // 2, ws // 2
        else:
            hin, win = hs, ws
        kh, kw = w.shape[2], w.shape[3]
        pt, _ = same_pad(hin, kh, stride[0])
        pl, _ = same_pad(win, kw, stride[1])
        grid = (math.ceil(hin / stride[0]), math.ceil(win / stride[1]))
        return self.conv(stage, name, srcs, w, self.sd[bkey] if bkey else None, out, stride=stride, pad=(pt, pl),
                         grid=grid, act=act, p0=p0, p1=p1, in_mode=in_mode)

    def conv_relu2(self, stage, name, srcs, prefix, mid, out, stride=1):
        """layers.ConvReLU2 (model/layers.py:308-314): k x 1 stride (s,1), then 1 x k stride (1,s)."""
        wy, wx = self.sd[prefix + ".conv_y.weight"], self.sd[prefix + ".conv_x.weight"]
        if (stride == 2 and self.winograd and self.conv_forms == "table" and self.bf16 == 0 and len(srcs) == 1 and
                name + ".conv_y" not in self.schedule_override and name + ".conv_x" not in self.schedule_override and
                srcs[0].shape[2] % 2 == 0 and srcs[0].shape[3] % 8 == 0 and tuple(wy.shape[2:]) == tuple(wx.shape[2:])[::-1]):
            code = choose_stride2(int(wy.shape[2]), int(wy.shape[0]), int(wy.shape[1]), int(out.shape[2]), int(out.shape[3]), int(out.shape[0]))
            if code:
                return self._conv_relu2_stride2(stage, name, srcs[0], prefix, mid, out, code // 10, code % 10)
        self.same_conv(stage, name + ".conv_y", srcs, prefix + ".conv_y.weight", prefix + ".conv_y.bias", mid,
                       stride=(stride, 1))
        return self.same_conv(stage, name + ".conv_x", [mid], prefix + ".conv_x.weight", prefix + ".conv_x.bias", out,
                              stride=(1, stride))

    def _conv_relu2_stride2(self, stage, name, x, prefix, mid, out, mbw_y, mbw_x):
        """A stride-2 layers.ConvReLU2 (model/layers.py:289-314; DepthModule.enc stages 1-3, monorec_model.py:489-501) on the stride-1 Cook-Toom
        kernel (csrc/conv1d_wino.hip, mr_conv1d_cooktoom_f32).  y_i = sum_k g_k d_{2i + k - pad} splits by the parity of the input sample into
        ONE ceil(k / 2)-tap stride-1 filter over the channel concatenation [even samples | odd samples] (cooktoom.stride2_as_stride1) - no data is
        moved for it:
          * k x 1, stride (2,1): the even / odd ROWS of x are two views with twice the row pitch (mr_wino_desc.src_row_pitch): F(4, ceil(k/2)) along y;
            its epilogue stores the result de-interleaved by COLUMN parity (dst_split_columns) - `mid` = (2, n, c, h/2, w/2) -,
          * 1 x k, stride (1,2): those two dense halves are its [even | odd] sources: F(4, ceil(k/2)) along x.
        7 taps -> F(4,4): 3.5 multiplies per output and input channel instead of 7; 5 taps -> F(4,3): 3 instead of 5."""
        n, c, hs, ws = [int(v) for v in x.shape]
        wy, by = self.sd[prefix + ".conv_y.weight"], self.sd[prefix + ".conv_y.bias"]
        wx, bx = self.sd[prefix + ".conv_x.weight"], self.sd[prefix + ".conv_x.bias"]
        k, cm, co = int(wy.shape[2]), int(wy.shape[0]), int(wx.shape[0])
        h2, w2 = hs // 2, ws // 2
        assert x.is_contiguous() and tuple(mid.shape) == (n, cm, h2, ws) and tuple(out.shape) == (n, co, h2, w2), (name, tuple(mid.shape), tuple(out.shape))
        sig = stride2_signature(k, cm, c, h2, w2, n)
        mid2 = mid.view(2, n, cm, h2, w2)
        uy = stride2_unified_weights(wy.detach().float().cpu(), hs, axis=1)
        view = dict(ptrs=[x.data_ptr(), x.data_ptr() + ws * 4], channels=[c, c], batch=n, height=h2, width=ws, row_pitch=2 * ws, plane=hs * ws)
        if mbw_x == 0:
            # only the k x 1 half on the Cook-Toom kernel (dense intermediate); the 1 x k half stays on the direct MFMA kernel with its tuned schedule -
            # what the table picks where the small x half is a latency chain (c2 depth.enc2.0: 17.9 + 22.4 us against 23.3 + 22.4)
            self._conv_winograd_1d(stage, name + ".conv_y", [x], uy, by, mid, ACT_LEAKY_RELU, LEAKY_SLOPE, 1, mbw_y, 4, view=view,
                                   ref_macs=n * h2 * ws * cm * c * k, ref_k=(k, 1), sig=sig + "_y")
            return self.same_conv(stage, name + ".conv_x", [mid], prefix + ".conv_x.weight", prefix + ".conv_x.bias", out, stride=(1, 2))
        self._conv_winograd_1d(stage, name + ".conv_y", [x], uy, by, mid2, ACT_LEAKY_RELU, LEAKY_SLOPE, 1, mbw_y, 4, view=view,
                               dst_split=True, ref_macs=n * h2 * ws * cm * c * k, ref_k=(k, 1), sig=sig + "_y")
        ux = stride2_unified_weights(wx.detach().float().cpu(), ws, axis=0)
        return self._conv_winograd_1d(stage, name + ".conv_x", [mid2[0], mid2[1]], ux, bx, out, ACT_LEAKY_RELU, LEAKY_SLOPE, 0, mbw_x, 4,
                                      ref_macs=n * h2 * w2 * co * cm * k, ref_k=(1, k), sig=sig + "_x")

    def refine(self, stage, name, srcs, prefix, out):
        """layers.Refine (model/layers.py:389-397): ConvTranspose2d(4, 2) + LeakyReLU + crop = 4 output-parity
        2x2 convolutions, run as the 4 phases of ONE launch."""
        wt = self.sd[prefix + ".conv2d_t.weight"]
        bias = self.sd[prefix + ".conv2d_t.bias"]
        h, w = srcs[0].shape[2], srcs[0].shape[3]
        if self.winograd and self.bf16 == 0 and name not in self.schedule_override and tuple(out.shape[2:]) == (2 * h, 2 * w):
            code = choose_winograd_t(int(wt.shape[1]), [int(s_.shape[1]) for s_ in srcs], h, w, int(srcs[0].shape[0]))
            if code:
                return self._refine_winograd(stage, name, srcs, wt, bias, out, code % 10, code // 10)
        phases = [(wp, pt, pl, py, px) for (py, px), (wp, pt, pl) in transposed_phase_weights(wt).items()]
        return self.conv(stage, name, srcs, None, bias, out, stride=(1, 1), grid=(h, w), act=ACT_LEAKY_RELU,
                         p0=LEAKY_SLOPE, out_step=(2, 2), phases=phases)

    def _refine_winograd(self, stage, name, srcs, wt, bias, out, mbw, variant):
        """One mr_convt4x4s2_winograd_f32 launch (csrc/convt_wino.hip: F(2x2,2x2), 9 instead of 16 multiplies per 2x2 parity tile) in
        place of the four-phase mr_conv2d_f32 launch of a Refine layer."""
        lib = self.lib
        n, _, hs, ws = srcs[0].shape
        src_channels = [int(s_.shape[1]) for s_ in srcs]
        cin, cout = int(wt.shape[0]), int(wt.shape[1])
        assert cin == sum(src_channels) and tuple(wt.shape[2:]) == (4, 4), (name, wt.shape, src_channels)
        sc = (ctypes.c_int32 * len(src_channels))(*src_channels)
        w = wt.detach().to(torch.float32).contiguous().cpu()
        if variant == 2:       # 32 a + (1..16) output channels: the tail group by 16-row workgroups (csrc/convt_wino.hip: convt_rb_tail)
            assert mbw == 1 and 0 < cout % 32 <= 16, (name, cout)
            nfl = lib.mr_wino_t_packed_weight_floats_tail(cout, sc, len(src_channels))
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino_t_pack_weights_tail_f32(w.data_ptr(), cout, sc, len(src_channels), packed.data_ptr()), "mr_wino_t_pack_weights_tail_f32")
        else:
            nfl = lib.mr_wino_t_packed_weight_floats(cout, sc, len(src_channels), mbw)
            packed = torch.empty(nfl, dtype=torch.float32)
            _lib.check(lib.mr_wino_t_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), mbw, packed.data_ptr()), "mr_wino_t_pack_weights_f32")
        d = WinoDesc()
        for i, s_ in enumerate(srcs):
            assert s_.is_contiguous() and tuple(s_.shape[2:]) == (hs, ws) and s_.shape[0] == n
            d.src[i], d.src_channels[i] = s_.data_ptr(), src_channels[i]
        d.num_src, d.batch, d.height, d.width = len(srcs), n, hs, ws
        assert out.is_contiguous() and tuple(out.shape) == (n, cout, 2 * hs, 2 * ws)
        d.dst, d.out_channels = out.data_ptr(), cout
        d.packed_weights = self._dev(packed).data_ptr()
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        d.activation, d.act_p0, d.cout_blocks_per_wave, d.variant = ACT_LEAKY_RELU, LEAKY_SLOPE, mbw, variant
        lds = lib.mr_convt4x4s2_winograd_lds_bytes(ctypes.byref(d))
        if lds < 0:
            _lib.check(int(lds), f"plan {name} winograd-t")
        ref = n * hs * ws * cout * cin * 16
        wgs = math.ceil(hs / 8) * math.ceil(ws / 32) * n * 4 * math.ceil(cout / (32 * mbw))
        self.conv_log.append(dict(name=name, macs=ref * 9 // 16, ref_macs=ref, mb=mbw, nb=0, split_k=1, ck=8, waves=8, kws=0, wgs=wgs, lds=int(lds),
                                  cout=cout, cin=cin, k=(2, 2), out=(hs, ws), batch=n, phases=4, winograd=mbw, wino_variant=variant, bf16=0,
                                  sig="t_" + winograd_signature(cout, src_channels, hs, ws, n),
                                  spec=dict(src_shapes=[tuple(s_.shape) for s_ in srcs], w_shape=(cout, cin, 2, 2), stride=(1, 1), pad=(0, 0),
                                            grid=(hs, ws), in_mode=IN_DIRECT, tf=TF_NONE, act=ACT_LEAKY_RELU, p0=LEAKY_SLOPE, p1=0.0, residual=False,
                                            out_shape=tuple(out.shape), out_step=(2, 2), out_off=(0, 0), phases=None)))
        self.keep += [d, out] + list(srcs)

        def run(stream):
            _lib.check(lib.mr_convt4x4s2_winograd_f32(ctypes.byref(d), stream), name)
        run.native = (_lib.LAUNCH_WINO_T, d, 0)
        self.stages[stage].append((name, run))
        return out

    def upconv(self, stage, name, srcs, wkey, bkey, out):
        """layers.Upconv (model/layers.py:349-356) phase-decomposed on the low-resolution input: the 4 output parities as the 4
        phases of ONE launch with 1, 2, 2 and 4 taps (upconv_phase_weights) - 2.25 instead of 4 multiply-adds per output, the
        input staged by the direct dwordx4 LDS-DMA path instead of the upsampling dword reads."""
        h, w = srcs[0].shape[2], srcs[0].shape[3]
        cout, cin = self.sd[wkey].shape[:2]
        if self.winograd and self.bf16 == 0 and name not in self.schedule_override and w % 4 == 0 and tuple(out.shape[2:]) == (2 * h, 2 * w):
            ukey = "u_" + winograd_signature(int(cout), [int(s_.shape[1]) for s_ in srcs], h, w, int(srcs[0].shape[0]))
            mbw = WINOGRAD[ukey] if ukey in WINOGRAD else (nearest_form("u_", int(cout), int(cin), h * w, int(srcs[0].shape[0]),
                                                                        valid=lambda c: c <= math.ceil(int(cout) / 16)) or 0)
            if mbw:        # measured table (tools/bench_wino1d.py --emit): the 4-multiply kernel, 16 * mbw output channels per workgroup
                return self._upconv_winograd(stage, name, srcs, self.sd[wkey], self.sd[bkey] if bkey else None, out, mbw)
        phases = [(wp, 0, 0, py, px) for (py, px), wp in upconv_phase_weights(self.sd[wkey]).items()]
        return self.conv(stage, name, srcs, None, self.sd[bkey] if bkey else None, out, stride=(1, 1), grid=(h, w), act=ACT_NONE,
                         out_step=(2, 2), phases=phases, ref_macs=srcs[0].shape[0] * 4 * h * w * cout * cin * 4)

    def _upconv_winograd(self, stage, name, srcs, weight, bias, out, mbw):
        """One mr_upconv2x2_winograd_f32 launch (csrc/conv1d_wino.hip: 4 multiplies per 2x2 output block instead of the 9 of the four
        parity phases / the 16 of the reference) for a layers.Upconv."""
        lib = self.lib
        n, _, hs, ws = srcs[0].shape
        src_channels = [int(s_.shape[1]) for s_ in srcs]
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        assert tuple(weight.shape[2:]) == (2, 2) and cin == sum(src_channels), (name, weight.shape)
        sc = (ctypes.c_int32 * len(src_channels))(*src_channels)
        w = weight.detach().to(torch.float32).contiguous().cpu()
        nfl = lib.mr_wino1d_packed_weight_floats(cout, sc, len(src_channels), mbw)
        packed = torch.empty(nfl, dtype=torch.float32)
        _lib.check(lib.mr_upconv_pack_weights_f32(w.data_ptr(), cout, sc, len(src_channels), mbw, packed.data_ptr()), "mr_upconv_pack_weights_f32")
        d = WinoDesc()
        for i, s_ in enumerate(srcs):
            assert s_.is_contiguous() and tuple(s_.shape[2:]) == (hs, ws) and s_.shape[0] == n
            d.src[i], d.src_channels[i] = s_.data_ptr(), src_channels[i]
        d.num_src, d.batch, d.height, d.width = len(srcs), n, hs, ws
        assert out.is_contiguous() and tuple(out.shape) == (n, cout, 2 * hs, 2 * ws)
        d.dst, d.out_channels = out.data_ptr(), cout
        d.packed_weights = self._dev(packed).data_ptr()
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        d.activation, d.act_p0, d.cout_blocks_per_wave = ACT_NONE, 0.0, mbw
        lds = lib.mr_conv1d3_winograd_lds_bytes(ctypes.byref(d))
        if lds < 0:
            _lib.check(int(lds), f"plan {name} upconv-winograd")
        ref = n * 4 * hs * ws * cout * cin * 4
        wgs = math.ceil(hs / 8) * math.ceil(ws / 32) * n * math.ceil(cout / (16 * mbw))
        self.conv_log.append(dict(name=name, macs=ref // 4, ref_macs=ref, mb=mbw, nb=0, split_k=1, ck=8, waves=8, kws=0, wgs=wgs, lds=int(lds),
                                  cout=cout, cin=cin, k=(2, 2), out=(hs, ws), batch=n, phases=4, winograd=mbw, wino_variant=0, upconv=True, bf16=0,
                                  sig="u_" + winograd_signature(cout, src_channels, hs, ws, n),
                                  spec=dict(src_shapes=[tuple(s_.shape) for s_ in srcs], w_shape=(cout, cin, 2, 2), stride=(1, 1), pad=(0, 0),
                                            grid=(hs, ws), in_mode=IN_DIRECT, tf=TF_NONE, act=ACT_NONE, p0=0.0, p1=0.0, residual=False,
                                            out_shape=tuple(out.shape), out_step=(2, 2), out_off=(0, 0), phases=None)))
        self.keep += [d, out] + list(srcs)

        def run(stream):
            _lib.check(lib.mr_upconv2x2_winograd_f32(ctypes.byref(d), stream), name)
        run.native = (_lib.LAUNCH_UPCONV, d, 0)
        self.stages[stage].append((name, run))
        return out

    # ------------------------------------------------------------------ bf16 MFMA path with B8 activation storage (csrc/conv_b8.hip)
    def alloc_b8(self, name, n, c, h, w):
        """Activation in the channel-blocked bf16 layout: (n, ceil(c / 8), h, w, 8) bf16; `.b8_channels` carries c."""
        t = torch.empty(n, (c + 7) // 8, h, w, 8, dtype=torch.bfloat16, device=self.device)
        t.b8_channels = c
        self.buf[name] = t
        return t

    @staticmethod
    def _act_info(t):
        """(layout, batch, channels, h, w) of an activation tensor: B8 (5-D bf16) or dense fp32 NCHW."""
        if t.dtype == torch.bfloat16:
            return LAYOUT_BF16_B8, int(t.shape[0]), int(t.b8_channels), int(t.shape[2]), int(t.shape[3])
        return LAYOUT_F32_NCHW, int(t.shape[0]), int(t.shape[1]), int(t.shape[2]), int(t.shape[3])

    @staticmethod
    def b8_schedule(cout, out_h, out_w, kh, kw, sh, sw, batch, phases, f32_source=False):
        """(MB, NB, waves) of a mr_conv2d_b8 launch: as many of the output channels per workgroup as 4 blocks of 16 allow (the input tile
        is then read once), 8 waves x 2 pixel blocks (8 x 32 pixels) where that still gives every CU two workgroups, smaller tiles below;
        bounded by the 160 KB of LDS and two staged tile positions per thread (csrc/conv_b8.hip: derive8)."""
        cb16 = (cout + 15) // 16
        mb = cb16 if cb16 <= 4 else (4 if cb16 % 4 == 0 else (3 if cb16 % 3 == 0 else 4))
        groups = math.ceil(cb16 / mb)
        import os
        # 16 x 32 pixel tiles first for the 2-D filters (7 LDS fragment reads per 12 MFMAs instead of 5 per 6: the 3x3 layers of 512x1024
        # gain 7-13 %, the k x 1 / 1 x k layers lose 20-30 %: r04_s4); MR_B8_NB4=0: A/B aid.  (>= 3 output blocks with 4 pixel blocks spill
        # next to the fp32 staging registers.)
        first = ((8, 4),) if (os.environ.get("MR_B8_NB4", "1") != "0" and not (mb >= 3 and f32_source) and kh > 1 and kw > 1) else ()
        for waves, nb in first + ((8, 2), (4, 2), (4, 1)):
            th = waves * nb // 2
            ih, iw = (th - 1) * sh + kh, 31 * sw + kw
            plane = ih * iw
            wgs = math.ceil(out_h / th) * math.ceil(out_w / 32) * groups * batch * phases
            lds = 2 * (64 * plane + 1024 * kh * kw * mb)
            if lds <= 160 * 1024 and plane <= 2 * 64 * waves and (wgs >= (1024 if nb == 4 else 512) or (waves, nb) == (4, 1)):
                return mb, nb, waves
        for waves, nb in ((8, 2), (8, 1), (4, 2), (4, 1)):          # whatever launches
            th = waves * nb // 2
            plane = ((th - 1) * sh + kh) * (31 * sw + kw)
            for m in (mb, 2, 1):
                if 2 * (64 * plane + 1024 * kh * kw * m) <= 160 * 1024 and plane <= 2 * 64 * waves:
                    return m, nb, waves
        raise ValueError("no launchable B8 schedule")

    def conv_b8(self, stage, name, srcs, weight, bias, out, *, stride=(1, 1), pad=(0, 0), grid=None, act=ACT_NONE, p0=0.0,
                out_step=(1, 1), phases=None, ref_macs=None):
        """Append one mr_conv2d_b8 launch.  srcs: activations (B8 or dense fp32 NCHW) concatenated on channels; out: B8 or fp32 NCHW.
        phases: optional list of 4 (weight, pad_top, pad_left, out_off_h, out_off_w) run in one launch."""
        lib = self.lib
        infos = [self._act_info(s_) for s_ in srcs]
        n, hs, ws = infos[0][1], infos[0][3], infos[0][4]
        assert all(i[1] == n and i[3] == hs and i[4] == ws for i in infos) and all(s_.is_contiguous() for s_ in srcs), name
        src_channels = [i[2] for i in infos]
        plist = [(weight, pad[0], pad[1], 0, 0)] if phases is None else phases
        cout, cin = int(plist[0][0].shape[0]), int(plist[0][0].shape[1])
        assert cin == sum(src_channels), (name, cin, src_channels)
        kh, kw = max(p[0].shape[2] for p in plist), max(p[0].shape[3] for p in plist)
        out_h, out_w = grid
        olay, on, oc, oh, ow = self._act_info(out)
        assert on == n and oc == cout and out.is_contiguous(), (name, on, n, oc, cout)
        f32_source = any(i[0] == LAYOUT_F32_NCHW for i in infos)
        sig = f"b8_co{cout}_ci{'+'.join(str(c) for c in src_channels)}_k{kh}x{kw}_s{stride[0]}x{stride[1]}_o{out_h}x{out_w}_b{n}_p{len(plist)}"
        sched8 = tuple(self.schedule_override.get(name) or B8_SCHEDULES.get(sig + f"_f{int(f32_source)}") or
                       self.b8_schedule(cout, out_h, out_w, kh, kw, stride[0], stride[1], n, len(plist), f32_source=f32_source))
        mb, nb, waves = sched8[:3]
        d = B8ConvDesc()
        sc = (ctypes.c_int32 * len(srcs))(*src_channels)
        for i, s_ in enumerate(srcs):
            d.src[i], d.src_channels[i], d.src_layout[i] = s_.data_ptr(), src_channels[i], infos[i][0]
            if "keyframe" in self.buf and s_ is self.buf["keyframe"]:
                self._input_srcs.append((d, i, "keyframe"))
        d.num_src, d.batch, d.src_h, d.src_w = len(srcs), n, hs, ws
        d.kh, d.kw, d.stride_h, d.stride_w = kh, kw, stride[0], stride[1]
        d.out_h, d.out_w = out_h, out_w
        d.dst, d.dst_layout, d.out_channels = out.data_ptr(), olay, cout
        d.dst_plane_h, d.dst_plane_w, d.out_step_h, d.out_step_w = oh, ow, out_step[0], out_step[1]
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        d.activation, d.act_p0 = act, p0
        d.cout_blocks_per_wg, d.pixel_blocks_per_wave, d.waves_per_wg = mb, nb, waves
        d.num_phases = len(plist)
        for i, (wp, pt, pl, oh_, ow_) in enumerate(plist):
            wc = wp.detach().to(torch.float32).contiguous().cpu()
            pk, pkw = int(wc.shape[2]), int(wc.shape[3])
            nbytes = lib.mr_b8_packed_weight_bytes(cout, sc, len(srcs), pk, pkw, mb)
            assert nbytes > 0, (name, mb)
            packed = torch.empty(nbytes, dtype=torch.uint8)
            _lib.check(lib.mr_b8_pack_weights(wc.data_ptr(), cout, sc, len(srcs), pk, pkw, mb, packed.data_ptr()), "mr_b8_pack_weights")
            dev_w = packed.to(self.device)
            self.keep.append(dev_w)
            d.phase_weights[i] = dev_w.data_ptr()
            d.phase_kh[i], d.phase_kw[i] = pk, pkw
            d.phase_pad_top[i], d.phase_pad_left[i], d.phase_out_off_h[i], d.phase_out_off_w[i] = pt, pl, oh_, ow_
        lds = lib.mr_conv2d_b8_lds_bytes(ctypes.byref(d))
        if lds < 0:
            _lib.check(int(lds), f"plan {name} b8 sched={(mb, nb, waves)}")
        taps = sum(int(p[0].shape[2]) * int(p[0].shape[3]) for p in plist)
        macs = n * out_h * out_w * cout * cin * taps
        th = waves * nb // 2
        wgs = math.ceil(out_h / th) * math.ceil(out_w / 32) * math.ceil(((cout + 15) // 16) / mb) * n * len(plist)
        self.conv_log.append(dict(name=name, macs=macs, ref_macs=macs if ref_macs is None else ref_macs, mb=mb, nb=nb, split_k=1, ck=32, waves=waves, kws=0,
                                  wgs=wgs, lds=int(lds), cout=cout, cin=cin, k=(kh, kw), out=(out_h, out_w), batch=n, phases=len(plist), bf16=1, b8=True,
                                  sig=sig, f32_source=f32_source,
                                  spec=dict(src_shapes=[(i[1], i[2], i[3], i[4]) for i in infos], src_layouts=[i[0] for i in infos], w_shape=(cout, cin, kh, kw),
                                            stride=tuple(stride), pad=tuple(pad), grid=(out_h, out_w), act=act, p0=p0, out_layout=olay,
                                            out_step=tuple(out_step))))
        self.keep += [d, out, sc] + list(srcs)

        def run(stream):
            _lib.check(lib.mr_conv2d_b8(ctypes.byref(d), stream), name)
        run.native = (_lib.LAUNCH_CONV_B8, d, 0)
        self.stages[stage].append((name, run))
        return out

    def same_conv_b8(self, stage, name, srcs, wkey, bkey, out, *, stride=(1, 1), act=ACT_LEAKY_RELU, p0=LEAKY_SLOPE):
        """PadSameConv2d + Conv2d (+ activation), model/layers.py:241-252,329-335, on the B8 kernel."""
        w = self.sd[wkey]
        _, _, _, hs, ws = self._act_info(srcs[0])
        kh, kw = w.shape[2], w.shape[3]
        pt, _ = same_pad(hs, kh, stride[0])
        pl, _ = same_pad(ws, kw, stride[1])
        grid = (math.ceil(hs / stride[0]), math.ceil(ws / stride[1]))
        return self.conv_b8(stage, name, srcs, w, self.sd[bkey] if bkey else None, out, stride=stride, pad=(pt, pl), grid=grid, act=act, p0=p0)

    def conv_relu2_b8(self, stage, name, srcs, prefix, mid, out, stride=1):
        """layers.ConvReLU2 (model/layers.py:308-314): k x 1 stride (s,1), then 1 x k stride (1,s)."""
        self.same_conv_b8(stage, name + ".conv_y", srcs, prefix + ".conv_y.weight", prefix + ".conv_y.bias", mid, stride=(stride, 1))
        return self.same_conv_b8(stage, name + ".conv_x", [mid], prefix + ".conv_x.weight", prefix + ".conv_x.bias", out, stride=(1, stride))

    def refine_b8(self, stage, name, srcs, prefix, out):
        """layers.Refine (model/layers.py:389-397): the four output parities as the four phases of one launch."""
        wt = self.sd[prefix + ".conv2d_t.weight"]
        _, _, _, h, w = self._act_info(srcs[0])
        phases = [(wp, pt, pl, py, px) for (py, px), (wp, pt, pl) in transposed_phase_weights(wt).items()]
        return self.conv_b8(stage, name, srcs, None, self.sd[prefix + ".conv2d_t.bias"], out, grid=(h, w), act=ACT_LEAKY_RELU, p0=LEAKY_SLOPE,
                            out_step=(2, 2), phases=phases)

    def upconv_b8(self, stage, name, srcs, wkey, bkey, out):
        """layers.Upconv (model/layers.py:349-356) phase-decomposed on the low-resolution input (upconv_phase_weights)."""
        _, n, _, h, w = self._act_info(srcs[0])
        cout, cin = self.sd[wkey].shape[:2]
        phases = [(wp, 0, 0, py, px) for (py, px), wp in upconv_phase_weights(self.sd[wkey]).items()]
        return self.conv_b8(stage, name, srcs, None, self.sd[bkey] if bkey else None, out, grid=(h, w), act=ACT_NONE, out_step=(2, 2), phases=phases,
                            ref_macs=n * 4 * h * w * cout * cin * 4)

    def add(self, stage, name, fn):
        self.stages[stage].append((name, fn))

    # ------------------------------------------------------------------ network
    def _build(self):
        B, H, W, F, D = self.B, self.H, self.W, self.F, self.D
        sd, lib = self.sd, self.lib
        kf = self.alloc("keyframe", B, 3, H, W)
        frames = self.alloc("frames", F, B, 3, H, W)
        geom = self.alloc("geom", B * 9 + B * F * 12)      # [kinv | proj], one H2D copy per forward
        self.buf["kinv"] = geom[: B * 9].view(B, 9)
        self.buf["proj"] = geom[B * 9:].view(B, F, 12)
        self.alloc("depths", D)

        # ---------------- ResNet-18 encoder (monorec_model.py:118-129) ----------------
        enc = "_feature_extractor.encoder"
        st = "encoder"
        w, b = fold_batchnorm(sd[enc + ".conv1.weight"], sd, enc + ".bn1")
        f0 = self.alloc("feat0", B, 64, H // 2, W // 2)
        # input normalisation as its own 3 us pass so that the stem runs on the DMA-staged conv path
        kfn = self.alloc("keyframe_norm", B, 3, H, W)

        self.input_ptr["keyframe"] = kf.data_ptr()

        def run_norm(stream, dst=kfn):
            _lib.check(lib.mr_resnet_normalize_f32(self.input_ptr["keyframe"], dst.data_ptr(), B * 3 * H * W, stream),
                       "mr_resnet_normalize_f32")
        self.add(st, "resnet.normalize", run_norm)
        self.conv(st, "resnet.conv1", [kfn], w, b, f0, stride=(2, 2), pad=(3, 3), grid=(H // 2, W // 2), act=ACT_RELU)
        pool = self.alloc("resnet.pool", B, 64, H // 4, W // 4)

        def run_pool(stream, src=self.ref(f0), dst=pool):
            _lib.check(lib.mr_maxpool3x3s2_f32(src.ptr(), dst.data_ptr(), B * 64, H // 2, W // 2, stream),
                       "mr_maxpool3x3s2_f32")
        self.add(st, "resnet.maxpool", run_pool)
        feats = [f0]
        x, cin, hh, ww = pool, 64, H // 4, W // 4
        for li, cout in enumerate((64, 128, 256, 512), start=1):
            if li == 4:
                if self.skip_layer4:
                    break                # dead work dropped on request: image_features[4] is absent from the output
                st = "encoder_tail"      # layer4 only feeds image_features[4]: nothing downstream waits for it
            for bi in range(2):
                pre = f"{enc}.layer{li}.{bi}"
                stride = 2 if (li > 1 and bi == 0) else 1
                ho, wo = hh // stride, ww // stride
                w1, b1 = fold_batchnorm(sd[pre + ".conv1.weight"], sd, pre + ".bn1")
                w2, b2 = fold_batchnorm(sd[pre + ".conv2.weight"], sd, pre + ".bn2")
                t = self.alloc(f"resnet.l{li}b{bi}.t", B, cout, ho, wo)
                self.conv(st, f"resnet.l{li}b{bi}.conv1", [x], w1, b1, t, stride=(stride, stride), pad=(1, 1),
                          grid=(ho, wo), act=ACT_RELU)
                idt = x
                if (pre + ".downsample.0.weight") in sd:
                    wd, bd = fold_batchnorm(sd[pre + ".downsample.0.weight"], sd, pre + ".downsample.1")
                    idt = self.alloc(f"resnet.l{li}b{bi}.idt", B, cout, ho, wo)
                    self.conv(st, f"resnet.l{li}b{bi}.down", [x], wd, bd, idt, stride=(stride, stride), pad=(0, 0),
                              grid=(ho, wo), act=ACT_NONE)
                is_feat = bi == 1
                o = self.alloc(f"feat{li}" if is_feat else f"resnet.l{li}b{bi}.o", B, cout, ho, wo)
                self.conv(st, f"resnet.l{li}b{bi}.conv2", [t], w2, b2, o, stride=(1, 1), pad=(1, 1), grid=(ho, wo),
                          act=ACT_RELU, residual=idt)
                x, cin, hh, ww = o, cout, ho, wo
            feats.append(x)

        # ---------------- cost volume (monorec_model.py:193-271) ----------------
        st = "cv"
        sfcv = self.alloc("sfcv", F, B, D, H, W)
        cv = self.alloc("cost_volume", B, D, H, W)
        frame_ptrs = (ctypes.c_void_p * F)(*[frames[f].data_ptr() for f in range(F)])
        sfcv_ptrs = (ctypes.c_void_p * F)(*[sfcv[f].data_ptr() for f in range(F)])
        self.keep += [frame_ptrs, sfcv_ptrs]
        self._ptr_arrays.append(sfcv_ptrs)
        self._frame_ptrs = frame_ptrs
        kinv, proj, depths = self.buf["kinv"], self.buf["proj"], self.buf["depths"]
        cv_ref = self.ref(cv)

        self._sfcv_b8_ptrs = None          # bf16 MFMA mode: pointer array of the B8 copies of the single-frame volumes (see _build_mask_depth_b8)

        def run_cv(stream):
            pix = self.buf["pix_depths"].data_ptr() if self.pix_depths_on else None     # data_dict["cv_depths"], :181-182
            if self._sfcv_b8_ptrs is not None and self.lean_outputs:
                _lib.check(lib.mr_cost_volume_b8_lean_f32(self.input_ptr["keyframe"], frame_ptrs, F, kinv.data_ptr(), proj.data_ptr(),
                                                          depths.data_ptr(), B, D, H, W, self.alpha, self.cw, self.cv_mode, pix,
                                                          cv_ref.ptr(), sfcv_ptrs, self._sfcv_b8_ptrs, stream), "mr_cost_volume_b8_lean_f32")
            elif self._sfcv_b8_ptrs is not None:
                _lib.check(lib.mr_cost_volume_b8_f32(self.input_ptr["keyframe"], frame_ptrs, F, kinv.data_ptr(), proj.data_ptr(),
                                                     depths.data_ptr(), B, D, H, W, self.alpha, self.cw, self.cv_mode, pix,
                                                     cv_ref.ptr(), sfcv_ptrs, self._sfcv_b8_ptrs, stream), "mr_cost_volume_b8_f32")
            elif self.cv_patch_size == 3 and self.cv_separable and self.sfcv_mult_mask:
                _lib.check(lib.mr_cost_volume_relaxed_f32(self.input_ptr["keyframe"], frame_ptrs, F, kinv.data_ptr(), proj.data_ptr(),
                                                          depths.data_ptr(), B, D, H, W, self.alpha, self.cw, self.cv_mode, pix,
                                                          cv_ref.ptr(), sfcv_ptrs, stream), "mr_cost_volume_relaxed_f32")
            elif self.cv_patch_size == 3:
                _lib.check(lib.mr_cost_volume_mode_f32(self.input_ptr["keyframe"], frame_ptrs, F, kinv.data_ptr(), proj.data_ptr(),
                                                       depths.data_ptr(), B, D, H, W, self.alpha, self.cw, self.cv_mode, pix,
                                                       1 if self.sfcv_mult_mask else 0,
                                                       cv_ref.ptr(), sfcv_ptrs, stream), "mr_cost_volume_mode_f32")
            else:
                _lib.check(lib.mr_cost_volume_patch_f32(self.input_ptr["keyframe"], frame_ptrs, F, kinv.data_ptr(), proj.data_ptr(),
                                                        depths.data_ptr(), B, D, H, W, self.alpha, self.cw, self.cv_mode, pix,
                                                        1 if self.sfcv_mult_mask else 0, self.cv_patch_size,
                                                        cv_ref.ptr(), sfcv_ptrs, stream), "mr_cost_volume_patch_f32")
        if self.no_cv:                     # :682-686: zero volumes, never written (the in-place mask multiply keeps 0)
            sfcv.zero_()
            cv.zero_()
        else:
            self.add(st, "cost_volume", run_cv)
        with_mask = self.pretrain_mode in (0, 2)
        with_depth = self.pretrain_mode != 2
        self.feats = feats
        if (self.b8 and self.pretrain_mode == 0 and not self.no_cv and not self.simple_mask and self.mask_use_cv and self.mask_use_feats and
                self.one_channel_kernels):
            return self._build_mask_depth_b8(feats, sfcv, cv, kf)
        self.b8 = False                    # the option variants keep the fp32-storage path

        # ---------------- MaskModule (monorec_model.py:345-385), frames batched as F*B ----------------
        am = "att_module"
        enc_ch = (int(sd[f"{am}.enc.0.0.conv.weight"].shape[1]) if with_mask else D, 48, 64, 96, 96)   # D; D + 4 for SimpleMaskModule
        FM = F                             # frames the mask encoder runs on
        x = sfcv.view(F * B, D, H, W)
        first_srcs = None
        if self.simple_mask and with_mask:
            # :447-453: one encoder pass over cat(non-zero mean of the single-frame volumes, keyframe, previous inverse depth)
            mean = self.alloc("mask.sfcv_mean", B, D, H, W)
            prev = self.alloc("prev_depth", B, 1, H, W)      # data_dict["predicted_inverse_depths"][0], copied in by the model

            def run_mean(stream, src=self.ref(sfcv), dst=mean):
                _lib.check(lib.mr_nonzero_mean_over_frames_f32(src.ptr(), dst.data_ptr(), self.F, B * D * H * W, stream),
                           "mr_nonzero_mean_over_frames_f32")
            self.add(st, "mask.sfcv_mean", run_mean)
            first_srcs, FM = [mean, kf, prev], 1
        elif not self.mask_use_cv:           # :352-353 `sfcv * 0`: every frame is the same zero input, so one pass stands for all
            x = self.alloc("mask.zero_input", B, D, H, W).zero_()
            FM = 1
        cvf = []
        for i in range(5 if with_mask else 0):
            hi, wi = H >> i, W >> i
            i0, i1 = (0, 1) if i == 0 else (1, 2)       # index 0 of stages 1-4 is the MaxPool
            a = self.alloc(f"mask.enc{i}.a", FM * B, enc_ch[i], hi, wi)
            xo = self.alloc(f"mask.enc{i}.x", FM * B, enc_ch[i], hi, wi)
            self.same_conv(st, f"mask.enc{i}.0", first_srcs if (i == 0 and first_srcs) else [x], f"{am}.enc.{i}.{i0}.conv.weight",
                           f"{am}.enc.{i}.{i0}.conv.bias", a)
            self.same_conv(st, f"mask.enc{i}.1", [a], f"{am}.enc.{i}.{i1}.conv.weight", f"{am}.enc.{i}.{i1}.conv.bias", xo)
            m = self.alloc(f"mask.cvf{i}", B, enc_ch[i], hi, wi)
            cvf.append(m)
            if i < 4:
                # nn.MaxPool2d(2) of the next stage (its own HBM-bound pass: the conv then stages by LDS-DMA) and the
                # maximum over the frames, both from one read of this stage's output
                xp = self.alloc(f"mask.enc{i + 1}.pool", FM * B, enc_ch[i], hi // 2, wi // 2)

                def run_pool_max(stream, src=xo, dst=xp, mx=m, planes=B * enc_ch[i], hh_=hi, ww_=wi):
                    _lib.check(lib.mr_pool2x2_framemax_f32(src.data_ptr(), dst.data_ptr(), mx.data_ptr(), FM, planes, hh_, ww_,
                                                           stream), "mr_pool2x2_framemax_f32")
                self.add(st, f"mask.poolmax{i}", run_pool_max)
                x = xp
            else:
                def run_max(stream, src=xo, dst=m, count=B * enc_ch[i] * hi * wi):
                    _lib.check(lib.mr_max_over_frames_f32(src.data_ptr(), dst.data_ptr(), FM, count, stream),
                               "mr_max_over_frames_f32")
                self.add(st, f"mask.max{i}", run_max)
        st = "main"
        mfeats = feats if self.mask_use_feats else [self.alloc(f"mask.zero_feat{i}", *feats[i].shape).zero_() for i in range(4)]   # :354-355
        x_srcs = [cvf[4], mfeats[3]] if with_mask else None                          # :372
        for i in range(4 if with_mask else 0):
            hi, wi = H >> (3 - i), W >> (3 - i)
            up_ch = sd[f"{am}.dec.{i}.0.conv.weight"].shape[0]
            dec_ch = {i: sd[f"{am}.dec.{i}.1.conv.weight"].shape[0]}
            u = self.alloc(f"mask.dec{i}.up", B, up_ch, hi, wi)
            self.upconv(st, f"mask.dec{i}.0", x_srcs, f"{am}.dec.{i}.0.conv.weight", f"{am}.dec.{i}.0.conv.bias", u)   # layers.py:349-356
            cat = [cvf[3 - i], u] if i == 3 else [cvf[3 - i], mfeats[2 - i], u]      # :374-380
            a = self.alloc(f"mask.dec{i}.a", B, dec_ch[i], hi, wi)
            xo = self.alloc(f"mask.dec{i}.x", B, dec_ch[i], hi, wi)
            self.same_conv(st, f"mask.dec{i}.1", cat, f"{am}.dec.{i}.1.conv.weight", f"{am}.dec.{i}.1.conv.bias", a)
            self.same_conv(st, f"mask.dec{i}.2", [a], f"{am}.dec.{i}.2.conv.weight", f"{am}.dec.{i}.2.conv.bias", xo)
            x_srcs = [xo]
        cv_mask = self.alloc("cv_mask", B, 1, H, W)
        mask_applied = False
        if with_mask and self.one_channel_kernels:
            # classifier (1x1 conv -> 1 channel, sigmoid, :340-343) and - in the full model - the mask multiply of :713 in one
            # HBM-bound launch (csrc/heads.hip) instead of a 1-of-16-rows MFMA launch + mr_apply_mask_f32
            feat = x_srcs[0]
            cw_ = self._dev(sd[f"{am}.classifier.0.weight"].reshape(-1))
            cb_ = self._dev(sd[f"{am}.classifier.0.bias"].reshape(-1))
            mask_applied = with_depth and self.pretrain_mode == 0
            cvp = cv_ref if mask_applied else None
            mask_ref = self.ref(cv_mask)

            def run_classifier(stream, feat=feat, cw_=cw_, cb_=cb_, cvp=cvp):
                _lib.check(lib.mr_mask_classifier_f32(feat.data_ptr(), cw_.data_ptr(), cb_.data_ptr(), B, int(feat.shape[1]), H * W,
                                                      mask_ref.ptr(), cvp.ptr() if cvp is not None else None, D, stream),
                           "mr_mask_classifier_f32")
            self.add(st, "mask.classifier", run_classifier)
            self.aux_log.append(dict(name="mask.classifier", ref_macs=B * H * W * int(feat.shape[1])))
        elif with_mask:
            self.same_conv(st, "mask.classifier", x_srcs, f"{am}.classifier.0.weight", f"{am}.classifier.0.bias", cv_mask,
                           act=ACT_SIGMOID)
        else:
            cv_mask.zero_()                # pretrain_mode 1, eval (:708); pretrain_mode 3: the model copies data_dict["mvobj_mask"] in (:711)
        self.feats = feats
        self.preds = None
        if not with_depth:                 # pretrain_mode 2 (:693, :712, :723-724): mask only
            return

        mask_ref2 = self.ref(cv_mask)

        def run_mask(stream):                                                        # :713 (in place)
            _lib.check(lib.mr_apply_mask_f32(cv_ref.ptr(), mask_ref2.ptr(), cv_ref.ptr(), B, D, H * W, stream),
                       "mr_apply_mask_f32")
        if self.pretrain_mode != 1 and not mask_applied:        # (1 - 0) * cv == cv exactly
            self.add(st, "apply_mask", run_mask)

        # ---------------- DepthModule (monorec_model.py:526-557) ----------------
        dm = "depth_module"
        # (channels, kernel, stride); the widths come from the weights (depth_large_model widens stages 3, 4, :482-483)
        enc_spec = tuple((int(sd[f"{dm}.enc.{i}.0.conv_y.weight"].shape[0]), k, s)
                         for i, (k, s) in enumerate(((7, 1), (7, 2), (5, 2), (5, 2), (3, 2))))
        dch = [int(sd[f"{dm}.dec.0.conv2d_t.weight"].shape[1]), int(sd[f"{dm}.dec.1.0.conv2d_t.weight"].shape[1]),
               int(sd[f"{dm}.dec.2.0.conv2d_t.weight"].shape[1]), int(sd[f"{dm}.dec.3.conv2d_t.weight"].shape[1]),
               int(sd[f"{dm}.dec.4.0.conv_y.weight"].shape[0]), int(sd[f"{dm}.dec.4.2.weight"].shape[0])]
        x_srcs = [cv, kf]                                                            # :531
        dfe = []
        hh, ww = H, W
        for i, (ch, _, s) in enumerate(enc_spec):
            h2, w2 = hh // s, ww // s
            mid0 = self.alloc(f"depth.enc{i}.0.mid", B, ch, h2, ww)
            o0 = self.alloc(f"depth.enc{i}.0.out", B, ch, h2, w2)
            self.conv_relu2(st, f"depth.enc{i}.0", x_srcs, f"{dm}.enc.{i}.0", mid0, o0, stride=s)
            mid1 = self.alloc(f"depth.enc{i}.1.mid", B, ch, h2, w2)
            o1 = self.alloc(f"depth.enc{i}.1.out", B, ch, h2, w2)
            self.conv_relu2(st, f"depth.enc{i}.1", [o0], f"{dm}.enc.{i}.1", mid1, o1)
            dfe.append(o1)
            x_srcs, hh, ww = [o1], h2, w2
        lo, hi_ = self.inv_depth_min_max[1], self.inv_depth_min_max[0]
        preds = [None] * 4

        heads = []                         # (predictor index, input, output): launched together behind the decoder

        def head(idx, src, scale_slot):
            hh_, ww_ = src.shape[2], src.shape[3]
            p = self.alloc(f"pred{scale_slot}", B, 1, hh_, ww_)
            if self.one_channel_kernels:
                heads.append((idx, src, p))
            else:
                self.same_conv(st, f"depth.head{idx}", [src], f"{dm}.predictors.{idx}.1.weight", f"{dm}.predictors.{idx}.1.bias",
                               p, act=ACT_ABS_TANH_AFFINE, p0=lo, p1=hi_)            # :556 + :717
            preds[scale_slot] = p

        r0 = self.alloc("depth.dec0", B, dch[0], H // 8, W // 8)
        self.refine(st, "depth.dec0", [dfe[4]], f"{dm}.dec.0", r0)
        head(0, r0, 3)
        r1 = self.alloc("depth.dec1.t", B, dch[1], H // 4, W // 4)
        self.refine(st, "depth.dec1.0", [dfe[3], feats[2], r0], f"{dm}.dec.1.0", r1)  # :545
        m1 = self.alloc("depth.dec1.mid", B, dch[1], H // 4, W // 4)
        x1 = self.alloc("depth.dec1", B, dch[1], H // 4, W // 4)
        self.conv_relu2(st, "depth.dec1.1", [r1], f"{dm}.dec.1.1", m1, x1)
        head(1, x1, 2)
        r2 = self.alloc("depth.dec2.t", B, dch[2], H // 2, W // 2)
        self.refine(st, "depth.dec2.0", [dfe[2], feats[1], x1], f"{dm}.dec.2.0", r2)
        m2 = self.alloc("depth.dec2.mid", B, dch[2], H // 2, W // 2)
        x2 = self.alloc("depth.dec2", B, dch[2], H // 2, W // 2)
        self.conv_relu2(st, "depth.dec2.1", [r2], f"{dm}.dec.2.1", m2, x2)
        head(2, x2, 1)
        x3 = self.alloc("depth.dec3", B, dch[3], H, W)
        self.refine(st, "depth.dec3", [dfe[1], feats[0], x2], f"{dm}.dec.3", x3)
        m4 = self.alloc("depth.dec4.mid", B, dch[4], H, W)
        x4a = self.alloc("depth.dec4.a", B, dch[4], H, W)
        self.conv_relu2(st, "depth.dec4.0", [dfe[0], x3], f"{dm}.dec.4.0", m4, x4a)   # :543
        x4 = self.alloc("depth.dec4", B, dch[5], H, W)
        self.same_conv(st, "depth.dec4.2", [x4a], f"{dm}.dec.4.2.weight", f"{dm}.dec.4.2.bias", x4)
        head(3, x4, 0)
        if heads:
            # the four predictors (3x3 conv -> 1 channel, abs(tanh), inverse-depth affine; :520-523,554-557,716-717) in ONE
            # HBM-bound launch (csrc/heads.hip): nothing downstream reads them, their inputs stay resident in the plan
            descs = (HeadDesc * len(heads))()
            for i, (idx, src, p) in enumerate(heads):
                w_ = self._dev(sd[f"{dm}.predictors.{idx}.1.weight"])
                b_ = self._dev(sd[f"{dm}.predictors.{idx}.1.bias"].reshape(-1))
                assert tuple(w_.shape) == (1, src.shape[1], 3, 3), w_.shape
                descs[i].src, descs[i].weight, descs[i].bias, descs[i].dst = src.data_ptr(), w_.data_ptr(), b_.data_ptr(), p.data_ptr()
                descs[i].batch, descs[i].channels, descs[i].height, descs[i].width = [int(v) for v in src.shape]
                self.keep += [src, p]
            self.keep.append(descs)

            def run_heads(stream, descs=descs, n=len(heads)):
                _lib.check(lib.mr_depth_heads_f32(descs, n, lo, hi_, stream), "mr_depth_heads_f32")
            self.add(st, "depth.heads", run_heads)
            self.aux_log.append(dict(name="depth.heads", ref_macs=sum(int(s_.shape[0] * s_.shape[1] * s_.shape[2] * s_.shape[3]) * 9
                                                                       for _, s_, _ in heads)))
        self.preds = preds

    def _build_mask_depth_b8(self, feats, sfcv, cv, kf):
        """MaskModule (monorec_model.py:345-385) and DepthModule (:526-557) of the full model in the bf16 MFMA mode with B8 activation
        storage (csrc/conv_b8.hip).  Dense fp32 stay: what the path hands out (single-frame / fused volumes, image features, `cv_mask`,
        the four depth scales) and the maps the HBM-bound one-channel kernels read (classifier input; the four head inputs)."""
        B, H, W, F, D = self.B, self.H, self.W, self.F, self.D
        sd, lib = self.sd, self.lib
        am, dm = "att_module", "depth_module"
        st = "cv"
        enc_ch = (int(sd[f"{am}.enc.0.0.conv.weight"].shape[0]), 48, 64, 96, 96)
        enc_ch = tuple(int(sd[f"{am}.enc.{i}.{0 if i == 0 else 1}.conv.weight"].shape[0]) for i in range(5))
        x = sfcv.view(F * B, D, H, W)
        if D in (32, 48, 64) and self.cv_patch_size == 3 and self.sfcv_mult_mask:
            # the fusion kernel of the cost volume writes every single-frame volume a second time in B8 (free: it is VALU-bound), so the
            # first mask-encoder layer reads 2 bytes per element by LDS-DMA instead of staging 4-byte planes through registers
            xb = self.alloc_b8("sfcv_b8", F * B, D, H, W)
            per_frame = B * (D // 8) * H * W * 16
            ptrs = (ctypes.c_void_p * F)(*[xb.data_ptr() + f * per_frame for f in range(F)])
            self.keep.append(ptrs)
            self._sfcv_b8_ptrs = ptrs
            x = xb
        cvf = []
        for i in range(5):
            hi, wi = H >> i, W >> i
            i0, i1 = (0, 1) if i == 0 else (1, 2)       # index 0 of stages 1-4 is the MaxPool
            a = self.alloc_b8(f"mask.enc{i}.a", F * B, enc_ch[i], hi, wi)
            xo = self.alloc_b8(f"mask.enc{i}.x", F * B, enc_ch[i], hi, wi)
            self.same_conv_b8(st, f"mask.enc{i}.0", [x], f"{am}.enc.{i}.{i0}.conv.weight", f"{am}.enc.{i}.{i0}.conv.bias", a)
            self.same_conv_b8(st, f"mask.enc{i}.1", [a], f"{am}.enc.{i}.{i1}.conv.weight", f"{am}.enc.{i}.{i1}.conv.bias", xo)
            m = self.alloc_b8(f"mask.cvf{i}", B, enc_ch[i], hi, wi)
            cvf.append(m)
            planes = B * ((enc_ch[i] + 7) // 8)
            if i < 4:
                xp = self.alloc_b8(f"mask.enc{i + 1}.pool", F * B, enc_ch[i], hi // 2, wi // 2)

                def run_pool_max(stream, src=xo, dst=xp, mx=m, planes=planes, hh_=hi, ww_=wi):
                    _lib.check(lib.mr_pool2x2_framemax_b8(src.data_ptr(), dst.data_ptr(), mx.data_ptr(), F, planes, hh_, ww_, stream),
                               "mr_pool2x2_framemax_b8")
                self.add(st, f"mask.poolmax{i}", run_pool_max)
                x = xp
            else:
                def run_max(stream, src=xo, dst=m, count=planes * hi * wi):
                    _lib.check(lib.mr_max_over_frames_b8(src.data_ptr(), dst.data_ptr(), F, count, stream), "mr_max_over_frames_b8")
                self.add(st, f"mask.max{i}", run_max)
        st = "main"
        # the image features (fp32 outputs of the ResNet trunk) are each read by TWO decoder layers: one conversion launch per feature map and
        # both readers take the B8 copy by LDS-DMA instead of staging 4-byte planes through registers (round 6: depth.dec3 157 us of which 148
        # are that staging path, profiles/r06_c5bf16_b8_ablation.txt).  Same round-to-nearest-even as the staging path: results bit-identical.
        fsrc = list(feats)
        if self.b8_feature_copies:
            for i in range(4):
                fb = self.alloc_b8(f"feat{i}_b8", B, int(feats[i].shape[1]), int(feats[i].shape[2]), int(feats[i].shape[3]))

                def run_fb(stream, src=self.ref(feats[i]), dst=fb, c_=int(feats[i].shape[1]), hw_=int(feats[i].shape[2] * feats[i].shape[3])):
                    _lib.check(lib.mr_f32_nchw_to_b8(src.ptr(), dst.data_ptr(), B, c_, hw_, stream), "mr_f32_nchw_to_b8")
                self.add(st, f"feat{i}.to_b8", run_fb)
                fsrc[i] = fb
        feats_out, feats = feats, fsrc
        x_srcs = [cvf[4], feats[3]]                                                  # :372
        for i in range(4):
            hi, wi = H >> (3 - i), W >> (3 - i)
            up_ch = int(sd[f"{am}.dec.{i}.0.conv.weight"].shape[0])
            dec_ch = int(sd[f"{am}.dec.{i}.1.conv.weight"].shape[0])
            u = self.alloc_b8(f"mask.dec{i}.up", B, up_ch, hi, wi)
            self.upconv_b8(st, f"mask.dec{i}.0", x_srcs, f"{am}.dec.{i}.0.conv.weight", f"{am}.dec.{i}.0.conv.bias", u)   # layers.py:349-356
            cat = [cvf[3 - i], u] if i == 3 else [cvf[3 - i], feats[2 - i], u]       # :374-380
            a = self.alloc_b8(f"mask.dec{i}.a", B, dec_ch, hi, wi)
            # the last decoder map feeds the classifier kernel (csrc/heads.hip), which reads dense fp32
            xo = self.alloc(f"mask.dec{i}.x", B, dec_ch, hi, wi) if i == 3 else self.alloc_b8(f"mask.dec{i}.x", B, dec_ch, hi, wi)
            self.same_conv_b8(st, f"mask.dec{i}.1", cat, f"{am}.dec.{i}.1.conv.weight", f"{am}.dec.{i}.1.conv.bias", a)
            self.same_conv_b8(st, f"mask.dec{i}.2", [a], f"{am}.dec.{i}.2.conv.weight", f"{am}.dec.{i}.2.conv.bias", xo)
            x_srcs = [xo]
        cv_mask = self.alloc("cv_mask", B, 1, H, W)
        feat = x_srcs[0]
        cw_ = self._dev(sd[f"{am}.classifier.0.weight"].reshape(-1))
        cb_ = self._dev(sd[f"{am}.classifier.0.bias"].reshape(-1))
        cv_ref, mask_ref = self.ref(cv), self.ref(cv_mask)
        cvb = self.alloc_b8("cost_volume_b8", B, D, H, W) if D % 16 == 0 else None     # the masked volume for the depth net, B8 copy

        def run_classifier(stream, feat=feat, cw_=cw_, cb_=cb_):
            if cvb is not None:
                _lib.check(lib.mr_mask_classifier_b8_f32(feat.data_ptr(), cw_.data_ptr(), cb_.data_ptr(), B, int(feat.shape[1]), H * W,
                                                         mask_ref.ptr(), cv_ref.ptr(), D, cvb.data_ptr(), stream), "mr_mask_classifier_b8_f32")
            else:
                _lib.check(lib.mr_mask_classifier_f32(feat.data_ptr(), cw_.data_ptr(), cb_.data_ptr(), B, int(feat.shape[1]), H * W,
                                                      mask_ref.ptr(), cv_ref.ptr(), D, stream), "mr_mask_classifier_f32")
        self.add(st, "mask.classifier", run_classifier)
        self.aux_log.append(dict(name="mask.classifier", ref_macs=B * H * W * int(feat.shape[1])))

        # ---------------- DepthModule ----------------
        enc_spec = tuple((int(sd[f"{dm}.enc.{i}.0.conv_y.weight"].shape[0]), k, s_)
                         for i, (k, s_) in enumerate(((7, 1), (7, 2), (5, 2), (5, 2), (3, 2))))
        dch = [int(sd[f"{dm}.dec.0.conv2d_t.weight"].shape[1]), int(sd[f"{dm}.dec.1.0.conv2d_t.weight"].shape[1]),
               int(sd[f"{dm}.dec.2.0.conv2d_t.weight"].shape[1]), int(sd[f"{dm}.dec.3.conv2d_t.weight"].shape[1]),
               int(sd[f"{dm}.dec.4.0.conv_y.weight"].shape[0]), int(sd[f"{dm}.dec.4.2.weight"].shape[0])]
        x_srcs = [cv if cvb is None else cvb, kf]                                    # :531
        dfe = []
        hh, ww = H, W
        for i, (ch, _, s_) in enumerate(enc_spec):
            h2, w2 = hh // s_, ww // s_
            mid0 = self.alloc_b8(f"depth.enc{i}.0.mid", B, ch, h2, ww)
            o0 = self.alloc_b8(f"depth.enc{i}.0.out", B, ch, h2, w2)
            self.conv_relu2_b8(st, f"depth.enc{i}.0", x_srcs, f"{dm}.enc.{i}.0", mid0, o0, stride=s_)
            mid1 = self.alloc_b8(f"depth.enc{i}.1.mid", B, ch, h2, w2)
            o1 = self.alloc_b8(f"depth.enc{i}.1.out", B, ch, h2, w2)
            self.conv_relu2_b8(st, f"depth.enc{i}.1", [o0], f"{dm}.enc.{i}.1", mid1, o1)
            dfe.append(o1)
            x_srcs, hh, ww = [o1], h2, w2
        lo, hi_ = self.inv_depth_min_max[1], self.inv_depth_min_max[0]
        # the maps the depth heads read (csrc/heads.hip) stay dense fp32: r0, x1, x2, x4
        r0 = self.alloc("depth.dec0", B, dch[0], H // 8, W // 8)
        self.refine_b8(st, "depth.dec0", [dfe[4]], f"{dm}.dec.0", r0)
        r1 = self.alloc_b8("depth.dec1.t", B, dch[1], H // 4, W // 4)
        self.refine_b8(st, "depth.dec1.0", [dfe[3], feats[2], r0], f"{dm}.dec.1.0", r1)  # :545
        m1 = self.alloc_b8("depth.dec1.mid", B, dch[1], H // 4, W // 4)
        x1 = self.alloc("depth.dec1", B, dch[1], H // 4, W // 4)
        self.conv_relu2_b8(st, "depth.dec1.1", [r1], f"{dm}.dec.1.1", m1, x1)
        r2 = self.alloc_b8("depth.dec2.t", B, dch[2], H // 2, W // 2)
        self.refine_b8(st, "depth.dec2.0", [dfe[2], feats[1], x1], f"{dm}.dec.2.0", r2)
        m2 = self.alloc_b8("depth.dec2.mid", B, dch[2], H // 2, W // 2)
        x2 = self.alloc("depth.dec2", B, dch[2], H // 2, W // 2)
        self.conv_relu2_b8(st, "depth.dec2.1", [r2], f"{dm}.dec.2.1", m2, x2)
        x3 = self.alloc_b8("depth.dec3", B, dch[3], H, W)
        self.refine_b8(st, "depth.dec3", [dfe[1], feats[0], x2], f"{dm}.dec.3", x3)
        m4 = self.alloc_b8("depth.dec4.mid", B, dch[4], H, W)
        x4a = self.alloc_b8("depth.dec4.a", B, dch[4], H, W)
        self.conv_relu2_b8(st, "depth.dec4.0", [dfe[0], x3], f"{dm}.dec.4.0", m4, x4a)   # :543
        x4 = self.alloc("depth.dec4", B, dch[5], H, W)
        self.same_conv_b8(st, "depth.dec4.2", [x4a], f"{dm}.dec.4.2.weight", f"{dm}.dec.4.2.bias", x4)
        heads = [(0, r0, 3), (1, x1, 2), (2, x2, 1), (3, x4, 0)]
        preds = [None] * 4
        descs = (HeadDesc * len(heads))()
        for i, (idx, src, slot) in enumerate(heads):
            p = self.alloc(f"pred{slot}", B, 1, int(src.shape[2]), int(src.shape[3]))
            preds[slot] = p
            w_ = self._dev(sd[f"{dm}.predictors.{idx}.1.weight"])
            b_ = self._dev(sd[f"{dm}.predictors.{idx}.1.bias"].reshape(-1))
            descs[i].src, descs[i].weight, descs[i].bias, descs[i].dst = src.data_ptr(), w_.data_ptr(), b_.data_ptr(), p.data_ptr()
            descs[i].batch, descs[i].channels, descs[i].height, descs[i].width = [int(v) for v in src.shape]
            self.keep += [src, p]
        self.keep.append(descs)

        def run_heads(stream, descs=descs, n=len(heads)):
            _lib.check(lib.mr_depth_heads_f32(descs, n, lo, hi_, stream), "mr_depth_heads_f32")
        self.add(st, "depth.heads", run_heads)
        self.aux_log.append(dict(name="depth.heads", ref_macs=sum(int(s_.shape[0] * s_.shape[1] * s_.shape[2] * s_.shape[3]) * 9 for _, s_, _ in heads)))
        self.preds = preds

    # ------------------------------------------------------------------ inputs
    def bind_inputs(self, keyframe, frames, in_place):
        """Point every launch that reads the keyframe / source frames at the caller's tensors (`in_place`: dense fp32 tensors of
        the plan's shape on its device - no device copy at all) or at the resident buffers after copying into them (anything
        else, and always under hipGraph replay, whose captured launches keep their pointers).  Returns True when bound in place."""
        ok = in_place and all(t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and
                              tuple(t.shape) == (self.B, 3, self.H, self.W) for t in [keyframe] + list(frames))
        if ok:
            kptr, fptrs = keyframe.data_ptr(), [f.data_ptr() for f in frames]
        else:
            self.buf["keyframe"].copy_(keyframe)
            for f in range(self.F):
                self.buf["frames"][f].copy_(frames[f])
            kptr, fptrs = self.buf["keyframe"].data_ptr(), [self.buf["frames"][f].data_ptr() for f in range(self.F)]
        self.input_ptr["keyframe"] = kptr
        for f in range(self.F):
            self._frame_ptrs[f] = fptrs[f]
        for d, i, _ in self._input_srcs:
            d.src[i] = kptr
        return ok

    # ------------------------------------------------------------------ execution
    def _compile_stage(self, stage):
        """The launch list of a stage as the host executes it: consecutive convolution-type launches (every closure that carries a
        `.native` = (MR_LAUNCH_* kind, descriptor, arg)) are folded into ONE mr_run_launches call each - ~90 of the ~105 C-ABI calls of
        a keyframe; everything else stays a Python closure.  The descriptors are referenced, not copied (input pointers are re-bound
        between forwards)."""
        out, run, names = [], [], []

        def flush():
            if not run:
                return
            items = (_lib.LaunchItem * len(run))()
            for i, (kind, desc, arg) in enumerate(run):
                items[i].kind, items[i].arg, items[i].desc = kind, arg, ctypes.addressof(desc)
            out.append((items, len(run), tuple(names)))
            self.keep.append(items)
            run.clear()
            names.clear()
        for name, fn in self.stages[stage]:
            nat = getattr(fn, "native", None)
            if nat is None:
                flush()
                out.append(fn)
            else:
                run.append(nat)
                names.append(name)
        flush()
        self._compiled[stage] = (len(self.stages[stage]), out)
        return out

    def run_stage(self, stage, stream):
        comp = self._compiled.get(stage)
        steps = comp[1] if comp is not None and comp[0] == len(self.stages[stage]) else self._compile_stage(stage)
        for st in steps:
            if isinstance(st, tuple):
                items, n, names = st
                failed = ctypes.c_int32(-1)
                rc = self.lib.mr_run_launches(items, n, stream, ctypes.byref(failed))
                if rc != 0:
                    _lib.check(rc, names[failed.value] if 0 <= failed.value < n else "mr_run_launches")
            else:
                st(stream)

    def conv_macs(self):
        """Multiply-adds the launches execute."""
        return sum(c["macs"] for c in self.conv_log)

    def conv_ref_macs(self):
        """Multiply-adds of the reference's Conv2d / ConvTranspose2d calls (forward hooks): the algorithmic work.
        Differs from conv_macs() by the phase-decomposed Upconv layers (9 instead of 16 taps per 2x2 output block)."""
        return sum(c["ref_macs"] for c in self.conv_log)
