"""GPU (MI355X): the bf16 MFMA path with channel-blocked bf16 activation storage (csrc/conv_b8.hip; BASELINE configs[4]) through the C ABI,
against torch on the CPU: operands rounded to bf16 (nearest even) exactly where the kernel rounds them, fp32 / fp64 arithmetic in between."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from monorec_amd import _lib, engine
from monorec_amd._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, LAYOUT_BF16_B8, LAYOUT_F32_NCHW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def to_b8(x):
    """dense (N, C, H, W) fp32 -> (N, ceil(C/8), H, W, 8) bf16 (padded channels zero)."""
    n, c, h, w = x.shape
    cb = (c + 7) // 8
    p = torch.zeros(n, cb * 8, h, w, dtype=torch.float32)
    p[:, :c] = x
    t = p.view(n, cb, 8, h, w).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)
    t.b8_channels = c
    return t


def from_b8(t, c):
    n, cb, h, w, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(n, cb * 8, h, w)[:, :c].contiguous()


def bf(x):
    return x.to(torch.bfloat16).float()


def _act(x, act, p0):
    return {ACT_NONE: lambda v: v, ACT_RELU: F.relu, ACT_LEAKY_RELU: lambda v: F.leaky_relu(v, p0)}[act](x)


def _run_b8(plan, srcs_cpu, layouts, weight, bias, out_c, out_hw, out_layout, **kw):
    dsrcs = []
    for x, lay in zip(srcs_cpu, layouts):
        if lay == LAYOUT_BF16_B8:
            t = to_b8(x).to(DEV)
            t.b8_channels = x.shape[1]
        else:
            t = x.to(DEV)
        dsrcs.append(t)
    n = srcs_cpu[0].shape[0]
    if out_layout == LAYOUT_BF16_B8:
        out = torch.full((n, (out_c + 7) // 8, out_hw[0], out_hw[1], 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        out.b8_channels = out_c
    else:
        out = torch.full((n, out_c, out_hw[0], out_hw[1]), float("nan"), device=DEV)
    plan.conv_b8("main", "t", dsrcs, weight, bias, out, **kw)
    plan.finalize()
    plan.run_stage("main", _stream())
    torch.cuda.synchronize()
    got = from_b8(out.cpu(), out_c) if out_layout == LAYOUT_BF16_B8 else out.cpu()
    assert torch.isfinite(got).all()
    return got, out


def _check(got, ref, out_layout, tag):
    """`ref`: the UNROUNDED fp32 reference.  A B8 destination rounds it once to bf16: at most half a bf16 ulp = 2^-8 relative (comparing
    against a rounded reference instead would flag every value whose fp32 sum lands on the other side of a rounding boundary)."""
    scale = max(1.0, float(ref.abs().max()))
    if out_layout == LAYOUT_BF16_B8:
        err = (got - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 1e-4 * scale
        assert bool((err <= tol).all()), (tag, float((err - tol).max()))
    else:
        assert float((got - ref).abs().max()) <= 2e-5 * scale * math.sqrt(ref.shape[1]), (tag, float((got - ref).abs().max()))


# (source channels, source layouts [0 fp32 / 1 B8], cout, (H, W), batch, (kh, kw), stride, act, out layout, schedule or None)
B8_CASES = [
    ((48,), (1,), 48, (24, 64), 2, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, None),                 # mask.enc: B8 -> B8
    ((48,), (0,), 48, (24, 64), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, None),                 # mask.enc0.0: fp32 single-frame volume in
    ((32, 3), (0, 0), 48, (16, 96), 1, (7, 1), (1, 1), ACT_LEAKY_RELU, 1, None),             # depth.enc0.0.conv_y: cost volume + keyframe, both fp32
    ((96, 128, 96), (1, 0, 1), 96, (16, 32), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, None),    # mask.dec: B8 / fp32 features / B8, 320 channels
    ((48,), (1,), 64, (32, 64), 1, (7, 1), (2, 1), ACT_LEAKY_RELU, 1, None),                 # depth.enc1.0.conv_y: 7 x 1 stride (2, 1)
    ((64,), (1,), 64, (16, 128), 1, (1, 7), (1, 2), ACT_LEAKY_RELU, 1, None),                # depth.enc1.0.conv_x: 1 x 7 stride (1, 2)
    ((64,), (1,), 128, (32, 64), 2, (5, 1), (2, 1), ACT_LEAKY_RELU, 1, None),
    ((128,), (1,), 128, (16, 64), 1, (1, 5), (1, 2), ACT_LEAKY_RELU, 1, None),
    ((32,), (1,), 24, (20, 40), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 0, None),                 # depth.dec4.2: fp32 out, 24 channels, ragged tile rows / columns
    ((5, 11), (1, 0), 40, (13, 20), 3, (3, 3), (1, 1), ACT_RELU, 1, None),                   # ragged everything: C % 8, cout % 16, H / W % tile
    ((3,), (0,), 7, (9, 5), 1, (1, 3), (1, 1), ACT_NONE, 0, None),
    ((48,), (1,), 48, (24, 64), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, (1, 1, 4)),             # other schedules of the same layer
    ((48,), (1,), 48, (24, 64), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, (3, 4, 4)),
    ((48,), (1,), 48, (24, 64), 1, (3, 3), (1, 1), ACT_LEAKY_RELU, 1, (2, 1, 8)),
    ((256,), (1,), 256, (8, 32), 1, (3, 1), (1, 1), ACT_LEAKY_RELU, 1, (4, 1, 4)),            # 16 output blocks, 8 K chunks
]


@pytest.mark.parametrize("case", range(len(B8_CASES)))
def test_conv_b8_matches_torch_on_bf16_rounded_operands(hip_lib, case):
    srcs_c, lays, cout, (h, w), batch, (kh, kw), (sh, sw), act, olay, sched = B8_CASES[case]
    g = torch.Generator().manual_seed(300 + case)
    srcs = [torch.randn(batch, c, h, w, generator=g) for c in srcs_c]
    cin = sum(srcs_c)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * (1.0 / math.sqrt(kh * kw * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    pt, _ = engine.same_pad(h, kh, sh)
    pl, _ = engine.same_pad(w, kw, sw)
    pb = sh * (math.ceil(h / sh) - 1) + kh - h - pt
    pr = sw * (math.ceil(w / sw) - 1) + kw - w - pl
    x = F.pad(bf(torch.cat(srcs, 1)).double(), (pl, pr, pt, pb))
    ref = _act(F.conv2d(x, bf(wt).double(), bias.double(), stride=(sh, sw)), act, 0.1).float()
    oh, ow = math.ceil(h / sh), math.ceil(w / sw)
    assert tuple(ref.shape[2:]) == (oh, ow)
    plan = engine.Plan.bare(DEV, schedule_override={"t": sched} if sched else None, bf16=1)
    got, _ = _run_b8(plan, srcs, lays, wt, bias, cout, (oh, ow), olay, stride=(sh, sw), pad=(pt, pl), grid=(oh, ow), act=act, p0=0.1)
    _check(got, ref, olay, B8_CASES[case])
    log = plan.conv_log[0]
    assert log["b8"] and log["macs"] == batch * oh * ow * cout * cin * kh * kw and 0 < log["lds"] <= 160 * 1024


@pytest.mark.parametrize("layer", ["refine", "upconv"])
@pytest.mark.parametrize("olay", [0, 1])
def test_b8_transposed_and_upsampling_layers(hip_lib, layer, olay):
    """layers.Refine (ConvTranspose2d(4, 2) + LeakyReLU + crop, model/layers.py:389-397) and layers.Upconv (nearest x2 -> pad (0,1,0,1) ->
    conv 2x2, :349-356) as four-phase B8 launches, against the reference formulation on the CPU."""
    g = torch.Generator().manual_seed(41 if layer == "refine" else 42)
    n, (h, w) = 2, (12, 40)
    srcs_c, lays = ((24, 16, 40), (1, 0, 0)) if layer == "refine" else ((40, 24), (1, 0))
    cin, cout = sum(srcs_c), 48
    srcs = [torch.randn(n, c, h, w, generator=g) for c in srcs_c]
    x = bf(torch.cat(srcs, 1)).double()
    sd = {}
    if layer == "refine":
        wt = torch.randn(cin, cout, 4, 4, generator=g) * (1.0 / math.sqrt(4.0 * cin))
        bias = torch.randn(cout, generator=g) * 0.1
        ref = F.leaky_relu(F.conv_transpose2d(x, bf(wt).double(), bias.double(), stride=2), 0.1)[:, :, 1:-1, 1:-1].float()
        sd = {"p.conv2d_t.weight": wt, "p.conv2d_t.bias": bias}
    else:
        wt = torch.randn(cout, cin, 2, 2, generator=g) * (1.0 / math.sqrt(4.0 * cin))
        bias = torch.randn(cout, generator=g) * 0.1
        # the phase filters are sums of taps formed in fp64 and rounded once to bf16: compare against exactly that arithmetic
        ref = torch.zeros(n, cout, 2 * h, 2 * w, dtype=torch.float64)
        xp = F.pad(x, (0, 1, 0, 1))
        for (py, px), wp in engine.upconv_phase_weights(wt).items():
            ref[:, :, py::2, px::2] = F.conv2d(xp[:, :, :h + py, :w + px], bf(wp).double(), bias.double())
        ref = ref.float()
        exact = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (0, 1, 0, 1)), wt.double(), bias.double()).float()
        assert float((ref - exact).abs().max()) <= 3e-2 * float(exact.abs().max())          # the bf16 weights, nothing structural
        sd = {"p.weight": wt, "p.bias": bias}
    plan = engine.Plan.bare(DEV, state=sd, bf16=1)
    dsrcs = []
    for s_, lay in zip(srcs, lays):
        t = to_b8(s_).to(DEV) if lay else s_.to(DEV)
        if lay:
            t.b8_channels = s_.shape[1]
        dsrcs.append(t)
    if olay:
        out = torch.full((n, cout // 8, 2 * h, 2 * w, 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        out.b8_channels = cout
    else:
        out = torch.full((n, cout, 2 * h, 2 * w), float("nan"), device=DEV)
    if layer == "refine":
        plan.refine_b8("main", "t", dsrcs, "p", out)
    else:
        plan.upconv_b8("main", "t", dsrcs, "p.weight", "p.bias", out)
    plan.finalize()
    plan.run_stage("main", _stream())
    torch.cuda.synchronize()
    got = from_b8(out.cpu(), cout) if olay else out.cpu()
    assert torch.isfinite(got).all()
    _check(got, ref, olay, (layer, olay))


def test_b8_pool_framemax_and_layout_conversions(hip_lib):
    lib = hip_lib
    g = torch.Generator().manual_seed(7)
    frames, b, c, h, w = 3, 2, 44, 12, 20
    x = bf(torch.randn(frames * b, c, h, w, generator=g))
    xb = to_b8(x).to(DEV)
    cb = (c + 7) // 8
    pooled = torch.empty(frames * b, cb, h // 2, w // 2, 8, dtype=torch.bfloat16, device=DEV)
    fmax = torch.empty(b, cb, h, w, 8, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.mr_pool2x2_framemax_b8(xb.data_ptr(), pooled.data_ptr(), fmax.data_ptr(), frames, b * cb, h, w, _stream()), "pool")
    mx = torch.empty(b, cb, h, w, 8, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.mr_max_over_frames_b8(xb.data_ptr(), mx.data_ptr(), frames, b * cb * h * w, _stream()), "max")
    back = torch.empty(frames * b, c, h, w, device=DEV)
    _lib.check(lib.mr_b8_to_f32_nchw(xb.data_ptr(), back.data_ptr(), frames * b, c, h * w, _stream()), "b8->f32")
    again = torch.empty_like(xb)
    _lib.check(lib.mr_f32_nchw_to_b8(back.data_ptr(), again.data_ptr(), frames * b, c, h * w, _stream()), "f32->b8")
    torch.cuda.synchronize()
    assert torch.equal(from_b8(pooled.cpu(), c), F.max_pool2d(x, 2))
    want = x.view(frames, b, c, h, w).max(0)[0]
    assert torch.equal(from_b8(fmax.cpu(), c), want) and torch.equal(from_b8(mx.cpu(), c), want)
    assert torch.equal(back.cpu(), x) and torch.equal(again.cpu().view(torch.int16), xb.cpu().view(torch.int16))
    assert lib.mr_pool2x2_framemax_b8(xb.data_ptr(), pooled.data_ptr(), fmax.data_ptr(), frames, b * cb, 11, w, _stream()) == -1      # odd height


def test_b8_bad_arguments_are_rejected_without_a_launch(hip_lib):
    d = _lib.B8ConvDesc()
    assert hip_lib.mr_conv2d_b8(ctypes.byref(d), _stream()) == -1
    assert hip_lib.mr_conv2d_b8_lds_bytes(ctypes.byref(d)) == -1
    sc = (ctypes.c_int32 * 1)(16)
    assert hip_lib.mr_b8_packed_weight_bytes(16, sc, 1, 3, 3, 5) == 0 and hip_lib.mr_b8_packed_weight_bytes(16, sc, 1, 3, 3, 1) == 9 * 1024


def test_cost_volume_and_classifier_write_their_b8_copies(hip_lib):
    """mr_cost_volume_b8_f32 / mr_mask_classifier_b8_f32: the dense fp32 outputs next to the plain entry points' (classifier: bit-identical; cost
    volume: single-frame volumes within 1e-4, relaxed window sums), and the B8 copies are the entry point's fp32 values rounded to bf16 (nearest even) in the
    (batch, D / 8, H, W, 8) layout."""
    from monorec_amd import synth
    from monorec_amd.model import MonoRecModel, host_geometry, depth_hypotheses
    lib = hip_lib
    b, h, w, nf, d = 2, 64, 96, 2, 32
    batch = synth.make_batch(b, h, w, nf, seed=5)
    kinv, proj = host_geometry(batch["keyframe_intrinsics"], batch["keyframe_pose"], batch["intrinsics"], batch["poses"])
    kf = batch["keyframe"].to(DEV)
    frames = [f.to(DEV) for f in batch["frames"]]
    kinv_d, proj_d = kinv.to(DEV), proj.to(DEV)
    depths = depth_hypotheses((0.33, 0.0025), d).to(DEV)
    cw = (ctypes.c_float * 3)(5 / 32, 16 / 32, 11 / 32)
    fptr = (ctypes.c_void_p * nf)(*[f.data_ptr() for f in frames])
    outs = {}
    for tag in ("plain", "b8"):
        cv = torch.empty(b, d, h, w, device=DEV)
        sf = torch.empty(nf, b, d, h, w, device=DEV)
        sptr = (ctypes.c_void_p * nf)(*[sf[f].data_ptr() for f in range(nf)])
        if tag == "plain":
            _lib.check(lib.mr_cost_volume_mode_f32(kf.data_ptr(), fptr, nf, kinv_d.data_ptr(), proj_d.data_ptr(), depths.data_ptr(), b, d, h, w, 10.0, cw, 1, None, 1,
                                                   cv.data_ptr(), sptr, _stream()), "cv")
        else:
            sfb = torch.full((nf * b, d // 8, h, w, 8), float("nan"), dtype=torch.bfloat16, device=DEV)
            bptr = (ctypes.c_void_p * nf)(*[sfb.data_ptr() + f * b * (d // 8) * h * w * 16 for f in range(nf)])
            _lib.check(lib.mr_cost_volume_b8_f32(kf.data_ptr(), fptr, nf, kinv_d.data_ptr(), proj_d.data_ptr(), depths.data_ptr(), b, d, h, w, 10.0, cw, 1, None,
                                                 cv.data_ptr(), sptr, bptr, _stream()), "cv b8")
        torch.cuda.synchronize()
        outs[tag] = (cv.cpu(), sf.cpu())
    # the bf16 configuration's entry point forms its 3x3 sums separably and multiplies by fp32(1/9) (csrc/cost_volume.hip, march_finish RELAXED):
    # the volumes move by an ulp of a 9-term sum against the SSIM constants - <= 1e-4, no validity flip - instead of being bit-identical
    dcv, dsf = (outs["plain"][0] - outs["b8"][0]).abs(), (outs["plain"][1] - outs["b8"][1]).abs()
    assert float(dsf.max()) <= 1e-4 and torch.equal(outs["plain"][1] == 0, outs["b8"][1] == 0)        # single-frame volumes, validity
    # the fused volume divides by the sum of the frame weights: where that sum is small the difference of the sads is amplified (as between
    # the reference and any other summation order) - a small outlier budget, like the use_ssim fixtures of test_gpu_kernels.py
    assert float((dcv > 1e-4).float().mean()) <= 2e-3 and float(dcv.max()) <= 5e-3 and torch.equal(outs["plain"][0] == 0, outs["b8"][0] == 0), \
        (float(dcv.max()), float((dcv > 1e-4).float().mean()))
    assert float((outs["plain"][1] - outs["b8"][1]).abs().max()) > 0            # (it IS the relaxed variant that ran)
    want = outs["b8"][1].view(nf * b, d, h, w)
    assert torch.equal(from_b8(sfb.cpu(), d), bf(want))                           # the B8 copy = the entry point's own fp32 values, rounded to bf16
    # classifier + mask multiply
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(b, 48, h, w, generator=g).to(DEV)
    wgt, bias = (torch.randn(48, generator=g) * 0.2).to(DEV), torch.zeros(1, device=DEV)
    res = {}
    for tag in ("plain", "b8"):
        cv = outs["plain"][0].clone().to(DEV)
        mask = torch.empty(b, 1, h, w, device=DEV)
        if tag == "plain":
            _lib.check(lib.mr_mask_classifier_f32(feat.data_ptr(), wgt.data_ptr(), bias.data_ptr(), b, 48, h * w, mask.data_ptr(), cv.data_ptr(), d, _stream()), "cls")
        else:
            cvb = torch.full((b, d // 8, h, w, 8), float("nan"), dtype=torch.bfloat16, device=DEV)
            _lib.check(lib.mr_mask_classifier_b8_f32(feat.data_ptr(), wgt.data_ptr(), bias.data_ptr(), b, 48, h * w, mask.data_ptr(), cv.data_ptr(), d, cvb.data_ptr(),
                                                     _stream()), "cls b8")
        torch.cuda.synchronize()
        res[tag] = (cv.cpu(), mask.cpu())
    assert torch.equal(res["plain"][0], res["b8"][0]) and torch.equal(res["plain"][1], res["b8"][1])
    assert torch.equal(from_b8(cvb.cpu(), d), bf(res["plain"][0]))
    assert lib.mr_mask_classifier_b8_f32(feat.data_ptr(), wgt.data_ptr(), bias.data_ptr(), b, 48, h * w, mask.data_ptr(), cv.data_ptr(), 24, cvb.data_ptr(), _stream()) == -1
