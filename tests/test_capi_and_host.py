"""CPU: the C-ABI library loads and exports what include/monorec_hip.h declares (no GPU compute),
and the host logic around it (padding, ConvTranspose phases, BN folding, weight packing, pose algebra,
model surface) is right."""
import ctypes
import glob
import json
import math
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import GOLDEN
from monorec_amd import _lib, engine, synth
from monorec_amd.model import MonoRecModel, depth_hypotheses, host_geometry
from oracle import monorec_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "monorec_hip.h")


# ------------------------------------------------------------------------------------------ ABI
def test_library_exports_every_declared_symbol(hip_lib):
    text = open(HEADER).read()
    diag = "".join(re.findall(r"#ifdef MR_DIAGNOSTIC_LIBRARY\n(.*?)#endif", text, flags=re.S))     # exported by the diagnostic build only
    assert set(re.findall(r"\b(mr_[a-z0-9_]+)\s*\(", diag)) == set(_lib.DIAGNOSTIC_ABI)
    text = re.sub(r"#ifdef MR_DIAGNOSTIC_LIBRARY\n.*?#endif", "", text, flags=re.S)
    declared = set(re.findall(r"\b(mr_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.ABI), (declared ^ set(_lib.ABI))
    for name in declared:
        assert getattr(hip_lib, name) is not None
    assert hip_lib.mr_abi_version() == _lib.MR_ABI_VERSION == int(re.search(r"#define MR_ABI_VERSION (\d+)", text).group(1))
    assert b"LDS" in hip_lib.mr_error_string(-3)


def test_conv_desc_layout_matches_the_c_struct():
    fields = ", ".join(f'offsetof(mr_conv_desc, {n})' for n, _ in _lib.ConvDesc._fields_)
    src = f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{ size_t o[] = {{{fields}}};' \
          'printf("%zu", sizeof(mr_conv_desc)); for (unsigned i = 0; i < sizeof(o)/sizeof(o[0]); ++i) printf(" %zu", o[i]); return 0; }'
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(c, "w").write(src)
        subprocess.run(["gcc", c, "-o", exe], check=True)
        vals = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_lib.ConvDesc)
    assert vals[1:] == [getattr(_lib.ConvDesc, n).offset for n, _ in _lib.ConvDesc._fields_]


def test_wino_desc_layout_matches_the_c_struct():
    """mr_wino_desc grew three fields in ABI 18 (strided source views, column-split destination: the stride-2 layers): ctypes mirror == C layout."""
    fields = ", ".join(f'offsetof(mr_wino_desc, {n})' for n, _ in _lib.WinoDesc._fields_)
    src = f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{ size_t o[] = {{{fields}}};' \
          'printf("%zu", sizeof(mr_wino_desc)); for (unsigned i = 0; i < sizeof(o)/sizeof(o[0]); ++i) printf(" %zu", o[i]); return 0; }'
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(c, "w").write(src)
        subprocess.run(["gcc", c, "-o", exe], check=True)
        vals = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_lib.WinoDesc)
    assert vals[1:] == [getattr(_lib.WinoDesc, n).offset for n, _ in _lib.WinoDesc._fields_]
    assert [n for n, _ in _lib.WinoDesc._fields_][-3:] == ["src_row_pitch", "src_plane_floats", "dst_split_columns"]


def test_bad_arguments_are_reported_not_crashed(hip_lib):
    d = _lib.ConvDesc()
    assert hip_lib.mr_conv2d_lds_bytes(ctypes.byref(d)) == -1
    assert hip_lib.mr_conv2d_f32(ctypes.byref(d), None) == -1
    assert hip_lib.mr_max_over_frames_f32(None, None, 2, 16, None) == -1
    assert hip_lib.mr_cost_volume_f32(None, None, 2, None, None, None, 1, 32, 64, 64, 10.0, None, None, None, None) == -1


def test_exact_constant_division_verdicts(hip_lib):
    """mr_exact_const_division: the cost-volume kernels divide by W - 1 / H - 1 (layers.py:67-68) with q = fma(r, y, q0), q0 = a y,
    r = fma(-d, q0, a) only where that equals the correctly rounded quotient for every dividend.  Cross-checked here with numpy on a
    binade (float64 evaluates both FMAs exactly enough: a - d q0 is exact in double)."""
    f32 = np.float32
    for d in (511.0, 255.0, 1023.0, 95.0):
        assert hip_lib.mr_exact_const_division(d) == 1
        a = np.arange(0x4b000000, 0x4b800000, 5, dtype=np.uint32).view(f32)
        y = f32(1.0) / f32(d)
        q0 = (a * y).astype(f32)
        r = (a.astype(np.float64) - np.float64(d) * q0.astype(np.float64)).astype(f32)
        q = (q0.astype(np.float64) + r.astype(np.float64) * np.float64(y)).astype(f32)
        assert np.array_equal(q, (a / f32(d)).astype(f32))
    assert hip_lib.mr_exact_const_division(1.0) == 0 and hip_lib.mr_exact_const_division(float("inf")) == 0


# ------------------------------------------------------------------------------------------ host logic
@pytest.mark.parametrize("n,k,s", [(256, 7, 2), (256, 5, 2), (256, 3, 2), (512, 2, 1), (64, 3, 1), (9, 7, 2), (8, 1, 2)])
def test_same_pad_matches_reference_rule(n, k, s):
    lo, hi = engine.same_pad(n, k, s)
    total = s * (math.ceil(n / s) - 1) + k - n
    assert lo + hi == total and lo == total // 2 and hi - lo in (0, 1)
    assert (n + lo + hi - k) // s + 1 == math.ceil(n / s)


def test_transposed_conv_phase_decomposition():
    torch.manual_seed(0)
    x = torch.randn(2, 5, 6, 7)
    wt = torch.randn(5, 4, 4, 4)
    bias = torch.randn(4)
    want = F.conv_transpose2d(x, wt, bias, stride=2)[:, :, 1:-1, 1:-1]      # layers.Refine crop
    got = torch.zeros_like(want)
    for (py, px), (w, pt, pl) in engine.transposed_phase_weights(wt).items():
        xp = F.pad(x, [pl, 1 - pl, pt, 1 - pt])
        got[:, :, py::2, px::2] = F.conv2d(xp, w, bias)
    assert torch.allclose(got, want, atol=1e-5)


def test_upconv_phase_decomposition_equals_upsample_pad_conv():
    """layers.Upconv (model/layers.py:349-356) = nearest x2, pad (0,1,0,1), conv 2x2; engine.upconv_phase_weights runs it as four
    (1+py) x (1+px) convolutions of the low-resolution input (zero beyond the bottom / right edge = the pad)."""
    torch.manual_seed(1)
    x = torch.randn(2, 5, 6, 7)
    w = torch.randn(4, 5, 2, 2)
    bias = torch.randn(4)
    want = F.conv2d(F.pad(F.interpolate(x, scale_factor=2), [0, 1, 0, 1]), w, bias)
    got = torch.zeros_like(want)
    taps = 0
    for (py, px), wp in engine.upconv_phase_weights(w).items():
        assert tuple(wp.shape[2:]) == (1 + py, 1 + px)
        taps += wp.shape[2] * wp.shape[3]
        got[:, :, py::2, px::2] = F.conv2d(F.pad(x, [0, px, 0, py]), wp, bias)
    assert taps == 9                                        # 2.25 multiply-adds per output instead of 4
    assert torch.allclose(got, want, atol=1e-5)


def test_batchnorm_folding():
    torch.manual_seed(0)
    w = torch.randn(8, 3, 3, 3)
    sd = {"bn.weight": torch.rand(8) + .5, "bn.bias": torch.randn(8), "bn.running_mean": torch.randn(8),
          "bn.running_var": torch.rand(8) + .5}
    x = torch.randn(2, 3, 9, 9)
    want = F.batch_norm(F.conv2d(x, w), sd["bn.running_mean"], sd["bn.running_var"], sd["bn.weight"], sd["bn.bias"],
                        False, 0.0, 1e-5)
    wf, bf = engine.fold_batchnorm(w, sd, "bn")
    assert torch.allclose(F.conv2d(x, wf, bf), want, atol=1e-5)


def _emulate_kernel_k_loop(packed, srcs, cout, kh, kw, stride, pad, grid, mb, ck):
    """Python model of conv_mfma_kernel's K walk over the packed weight stream (conv_layout.h):
    per cout group g: for chunk, tap, c4, m: D[(g*mb+m)*16 + i][pixel] += A[i][k] * B[k][pixel]."""
    n = srcs[0].shape[0]
    ho, wo = grid
    cb_n = (cout + 15) // 16
    groups = (cb_n + mb - 1) // mb
    out = np.zeros((n, groups * mb * 16, ho, wo), np.float64)
    off = 0
    padded = []
    for s in srcs:
        c_real = s.shape[1]
        cpad = (c_real + 3) // 4 * 4
        xp = np.zeros((n, cpad, s.shape[2] + 16, s.shape[3] + 16), np.float64)
        xp[:, :c_real, 8:-8, 8:-8] = s.numpy()
        padded.append((xp, cpad))
    for g in range(groups):
        for xp, cpad in padded:
            for c0 in range(0, cpad, ck):
                ckq = min(ck, cpad - c0)
                for tap in range(kh * kw):
                    ky, kx = divmod(tap, kw)
                    ys = 8 - pad[0] + ky + stride[0] * np.arange(ho)
                    xs = 8 - pad[1] + kx + stride[1] * np.arange(wo)
                    for c4 in range(ckq // 4):
                        for m in range(mb):
                            a = packed[off:off + 64].numpy().astype(np.float64).reshape(4, 16)   # lane = k*16 + i
                            off += 64
                            cb = g * mb + m
                            for k in range(4):
                                b = xp[:, c0 + c4 * 4 + k][:, ys][:, :, xs]                      # (n, ho, wo)
                                out[:, cb * 16:(cb + 1) * 16] += a[k][None, :, None, None] * b[:, None]
    assert off == packed.numel()
    return torch.from_numpy(out[:, :cout]).float()


@pytest.mark.parametrize("srcs_c,cout,k,stride,pad,mb,ck", [
    ((32, 3), 48, (7, 1), (1, 1), (3, 0), 3, 16),      # DepthModule enc0 conv_y: concat of cv + keyframe (3 -> pad 4)
    ((3,), 64, (7, 7), (2, 2), (3, 3), 2, 16),         # ResNet stem
    ((24,), 1, (3, 3), (1, 1), (1, 1), 1, 16),         # head on 24 channels (16 + 8 chunk), single output channel
    ((16, 20, 8), 40, (2, 2), (1, 1), (1, 0), 2, 16),  # three sources, partial last cout group
    ((72, 40), 96, (3, 3), (1, 1), (1, 1), 4, 32),     # deeper chunks, group count not dividing the cout blocks
    ((128,), 32, (1, 3), (1, 2), (0, 1), 1, 64),
])
def test_packed_weight_stream_matches_conv2d(hip_lib, srcs_c, cout, k, stride, pad, mb, ck):
    torch.manual_seed(1)
    n, h, w = 1, 10, 12
    srcs = [torch.randn(n, c, h, w) for c in srcs_c]
    weight = torch.randn(cout, sum(srcs_c), *k)
    ho = (h + 2 * pad[0] - k[0]) // stride[0] + 1
    wo = (w + 2 * pad[1] - k[1]) // stride[1] + 1
    want = F.conv2d(F.pad(torch.cat(srcs, 1), [pad[1], pad[1], pad[0], pad[0]]), weight, stride=stride)
    packed = engine.pack_conv_weight(weight, list(srcs_c), mb, ck)
    got = _emulate_kernel_k_loop(packed, srcs, cout, k[0], k[1], stride, pad, (ho, wo), mb, ck)
    assert torch.allclose(got, want, atol=1e-4), float((got - want).abs().max())


def test_host_geometry_is_bit_identical_to_the_oracle_algebra():
    for hard in (False, True):
        b = synth.make_batch(2, 64, 96, 3, seed=5, hard_pose=hard)
        kinv, proj = host_geometry(b["keyframe_intrinsics"], b["keyframe_pose"], b["intrinsics"], b["poses"])
        for n in range(2):
            assert torch.equal(kinv[n].view(3, 3), torch.inverse(b["keyframe_intrinsics"][n])[:3, :3])
            for f in range(3):
                want = orc.projection_matrix(b["intrinsics"][f][n], b["poses"][f][n], b["keyframe_pose"][n])
                assert torch.equal(proj[n, f].view(3, 4), want[0])


def test_depth_hypotheses_equal_reference_formula():
    for d in (8, 32, 48, 64):
        assert torch.equal(depth_hypotheses((0.33, 0.0025), d), orc.depth_hypotheses(0.33, 0.0025, d))


# ------------------------------------------------------------------------------------------ model surface
def test_state_dict_keys_equal_reference_fixture():
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys_d32.json")))
    got = {k: list(v.shape) for k, v in MonoRecModel(cv_depth_steps=32).state_dict().items()}
    assert got == want


def test_public_attributes_are_json_serialisable():
    m = MonoRecModel(inv_depth_min_max=[0.33, 0.0025], checkpoint_location=None, pretrain_mode=0, pretrain_dropout=0,
                     use_stereo=False, use_mono=True, use_ssim=1)
    public = {k: v for k, v in m.__dict__.items() if not k.startswith("_")}     # evaluate.py:36-43
    json.dumps({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in public.items()})
    assert public["cv_depth_steps"] == 32 and public["training"] is True


@pytest.mark.parametrize("kw", [dict(use_mono=False), dict(pretrain_mode=4), dict(use_ssim=4),
                                dict(cv_patch_size=4), dict(cv_patch_size=9), dict(augmentation="depth")])
def test_unsupported_options_raise(kw):
    with pytest.raises(NotImplementedError):
        MonoRecModel(**kw)


def test_pretrain_modes_build_the_reference_submodules():
    """monorec_model.py:622-628: no MaskModule for pretrain_mode 1 / 3, no DepthModule for 2 (checkpoints of those stages load strictly)."""
    has = lambda m, prefix: any(k.startswith(prefix) for k in m.state_dict())
    for mode, mask, depth in ((0, True, True), (1, False, True), (2, True, False), (3, False, True)):
        m = MonoRecModel(cv_depth_steps=8, pretrain_mode=mode)
        assert has(m, "att_module.") == mask and has(m, "depth_module.") == depth, mode
        assert has(m, "_feature_extractor.")
    simple = MonoRecModel(cv_depth_steps=8, simple_mask=True).state_dict()          # SimpleMaskModule: depth_steps + 3 + 1 input channels
    assert simple["att_module.enc.0.0.conv.weight"].shape == (12, 12, 3, 3) and simple["att_module.enc.1.1.conv.weight"].shape[1] == 12


def test_forward_refuses_cpu_inputs_and_training_mode():
    m = MonoRecModel(cv_depth_steps=8)
    batch = synth.make_batch(1, 64, 96, 2)
    with pytest.raises(NotImplementedError):
        m(batch)                       # still in training mode
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(batch)
    with pytest.raises(KeyError):
        m({"keyframe": batch["keyframe"]})


def test_checkpoint_loading_strips_dataparallel_prefix(tmp_path):
    m = MonoRecModel(cv_depth_steps=8)
    sd = synth.seeded_state_dict(m.state_dict(), seed=3)
    cp = tmp_path / "cp.pth"
    torch.save({"arch": "DataParallel", "state_dict": {"module." + k: v for k, v in sd.items()}}, cp)
    m2 = MonoRecModel(cv_depth_steps=8, checkpoint_location=[str(cp)])
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)


def test_launch_stamp_follows_the_launches_of_the_plan_not_the_tables(hip_lib, monkeypatch):
    """Profile sets are stamped with Plan.launch_stamp() (bench.py quotes a committed rocprof figure only on an equal stamp): equal for
    equal launch lists, different as soon as one layer's schedule or kernel family differs, untouched by table entries of other shapes."""
    m = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(m.state_dict())
    p1 = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    p2 = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    assert p1.launch_stamp() == p2.launch_stamp() and len(p1.launch_stamp()) == 16
    monkeypatch.setitem(engine.TUNED, "co999_ci1_k3x3_s1x1_o8x8_b1_p1", (1, 1, 1, 8, 8, 0))      # a foreign shape
    assert engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu").launch_stamp() == p1.launch_stamp()
    direct = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", winograd=False)
    assert [c for c in p1.conv_log if c.get("winograd")] and direct.launch_stamp() != p1.launch_stamp()
    first = next(c for c in p1.conv_log if not c.get("winograd"))
    other = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu",
                        schedule_override={first["name"]: (first["mb"], first["nb"], first["split_k"], max(8, first["ck"] // 2) if first["ck"] > 8 else 16, first["waves"], first["kws"])})
    assert other.launch_stamp() != p1.launch_stamp()


def test_bench_quotes_a_profile_set_only_on_an_equal_stamp(tmp_path, monkeypatch):
    """bench.profile_is_current: the newest committed set of a workload is quoted only when its stamp equals the running plan's, and only
    when the stamp file belongs to the newest trace of that workload."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    prof = tmp_path / "profiles"
    prof.mkdir()
    assert bench.profile_is_current("c2", "abc")[0] is False                      # nothing committed
    (prof / "r09_c2_stamp.json").write_text(json.dumps({"plan_stamp": "abc"}))
    assert bench.profile_is_current("c2", "abc")[0] is False                      # a stamp without its traces
    (prof / "r09_c2_kernel_stats_seq.csv").write_text("name\n")
    ok, info = bench.profile_is_current("c2", "abc")
    assert ok and info["profile"] == "abc" and info["profile_file"].endswith("r09_c2_stamp.json")
    assert bench.profile_is_current("c2", "other")[0] is False                    # the plan moved on
    (prof / "r10_c2_kernel_stats_seq.csv").write_text("name\n")                   # a newer trace without a stamp of its own
    assert bench.profile_is_current("c2", "abc")[0] is False


# Round 6 is moving the ABI (18 -> 19) and the tables; the committed profile sets are regenerated on the final plan in the round's last hardware
# session (tools/sessions/r06_*).  Until then bench.py reports `stale_profile` and omits the kernel-only figures, as designed.  REMOVE when done.
PROFILES_PENDING_REGENERATION = False


def test_bench_counts_every_conv_and_splitk_kernel_of_the_committed_trace():
    """bench.committed_kernel_stats matched `splitk_epilogue_kernel` only and left the `splitk_epilogue4_kernel<KS>` finishers
    (60 us per c2 keyframe) out of `conv_us_per_forward`.  The rule is now "every kernel with `conv` in its name + every kernel with `splitk` in its
    name"; recompute it independently from every committed one-keyframe-at-a-time trace and require equality."""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    checked = 0
    for tag in ("c2", "c3", "c5bf16"):
        got, src = bench.committed_kernel_stats(tag)
        if got is None:
            continue
        rows = list(csv.reader(open(os.path.join(ROOT, src))))[1:]
        forwards = sum(int(r[1]) for r in rows if "cv_sad" in r[0])
        want = sum(float(r[2]) for r in rows if "conv" in r[0] or "splitk" in r[0]) / forwards
        assert abs(got["conv_us_per_forward"] - want) < 1e-6 * want, (tag, got["conv_us_per_forward"], want)
        fin = sum(float(r[2]) for r in rows if "splitk" in r[0]) / forwards
        assert abs(got["splitk_finish_us_per_forward"] - fin) < 1e-6 * max(fin, 1.0)
        assert any("splitk_epilogue4_kernel" in k for k in got["by_kernel"]) or fin == 0.0 or tag != "c2"
        checked += 1
    assert checked >= 1


def test_skip_dead_layer4_drops_exactly_the_layer4_launches(hip_lib):
    """MonoRecModel(hip_skip_dead_layer4=True) -> Plan(skip_layer4=True): ResNet layer4 (monorec_model.py:118-129; read by nobody, :372-380,545) is not
    launched, everything else is the same launch for launch."""
    m = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(m.state_dict())
    full = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    lean = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", skip_layer4=True)
    names = lambda p: [n for st in ("encoder", "encoder_tail", "cv", "main") for n, _ in p.stages[st]]
    dropped = [n for n in names(full) if n not in names(lean)]
    assert dropped == ["resnet.l4b0.conv1", "resnet.l4b0.down", "resnet.l4b0.conv2", "resnet.l4b1.conv1", "resnet.l4b1.conv2"]
    assert names(lean) == [n for n in names(full) if n not in dropped] and not lean.stages["encoder_tail"]
    assert len(lean.feats) == 4 and "feat4" not in lean.buf
    assert abs((full.conv_ref_macs() - lean.conv_ref_macs()) / 1e9 - 1.074) < 0.001          # layer4 = 1.07 GMAC at c1
    assert MonoRecModel(hip_skip_dead_layer4=True)._skip_layer4 is True and m._skip_layer4 is False


@pytest.mark.xfail(PROFILES_PENDING_REGENERATION, reason="profile sets of round 4 (ABI 17) until the final session of round 5 regenerates them", strict=False)
def test_committed_profile_sets_belong_to_the_plans_of_this_tree(hip_lib):
    """The profile sets under profiles/ that bench.py quotes (c2, c3, configs[4] bf16) carry the launch stamp of the plan THIS tree builds for
    their workload: a table or ABI change without regenerated profiles fails here instead of showing up as `stale_profile` on the driver's line."""
    for tag, (b, h, w, f, d, bf) in {"c2": (1, 256, 512, 2, 32, 0), "c3": (8, 256, 512, 4, 64, 0), "c5bf16": (1, 512, 1024, 4, 48, 1)}.items():
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{tag}_stamp.json")))
        assert files, tag
        m = MonoRecModel(cv_depth_steps=d)
        plan = engine.Plan(synth.seeded_state_dict(m.state_dict()), b, h, w, f, d, (0.33, 0.0025), "cpu", bf16=bf)
        assert json.load(open(files[-1]))["plan_stamp"] == plan.launch_stamp(), (tag, files[-1])


def test_plan_dry_run_on_cpu_accounts_for_every_mac(hip_lib):
    """Builds the whole launch plan with CPU buffers (no launches): every descriptor validates, LDS stays
    inside a CU, and the conv MACs equal the reference's hook count (61.07 GMAC @ c1)."""
    m = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(m.state_dict())
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    # the five one-channel layers (classifier + four depth heads, 0.068 GMAC) run on their own HBM-bound kernels (csrc/heads.hip)
    aux = sum(a["ref_macs"] for a in plan.aux_log)
    assert [a["name"] for a in plan.aux_log] == ["mask.classifier", "depth.heads"] and abs(aux / 1e9 - 0.068) < 0.001
    assert abs((plan.conv_ref_macs() + aux) / 1e9 - 61.07) < 0.01
    # executed: the four mask-decoder Upconv layers (3.934 GMAC in the reference) run phase-decomposed at 9/16 of their taps, and the
    # 3x3 layers the measured table sends to the Winograd kernel (csrc/conv_wino.hip) at 16/36 of their multiplies
    wino = [c for c in plan.conv_log if c.get("winograd") and c["phases"] == 1 and tuple(c["k"]) == (3, 3)]
    assert {c["name"] for c in wino} == {"mask.enc0.0", "mask.enc0.1", "mask.enc1.0", "mask.enc1.1", "mask.dec2.1", "mask.dec2.2", "mask.dec3.1",
                                         "mask.dec3.2", "depth.dec4.2"}
    # (F(2x2,3x3): 16 of 36 multiplies; F(4x4,3x3), csrc/conv_wino44.hip: 36 of 144 - the four full-resolution layers since r04_s26)
    f44 = {c["name"] for c in wino if c.get("wino_variant") == 3}
    assert f44 == {"mask.enc0.0", "mask.enc0.1", "mask.dec3.1", "mask.dec3.2"}
    assert all((c["macs"] * 4 == c["ref_macs"] if c["name"] in f44 else c["macs"] * 9 == c["ref_macs"] * 4) and c["lds"] <= 160 * 1024 for c in wino)
    assert not [c["name"] for c in wino if c.get("wino_variant") == 2]     # (the 48-channel layers ran 32 + a 16-channel tail on F(2x2,3x3))
    # ... and the two large Refine layers (ConvTranspose2d(4, 2)) on the F(2x2,2x2) kernel (csrc/convt_wino.hip) at 9/16
    wino_t = [c for c in plan.conv_log if c.get("winograd") and c["phases"] == 4 and not c.get("upconv")]
    assert {c["name"] for c in wino_t} == {"depth.dec2.0", "depth.dec3"}
    assert all(c["macs"] * 16 == c["ref_macs"] * 9 and c["lds"] <= 160 * 1024 and c["sig"].startswith("t_") for c in wino_t)
    # ... and twelve of the sixteen 3 x 1 / 1 x 3 stride-1 layers of the depth net plus the two 7-tap layers of enc.0.0 on the 1-D kernels
    # (csrc/conv1d_wino.hip): F(2,3) at 4/6, or the Cook-Toom form F(m, r) the table names at (m + r - 1) / (m r)
    wino_1d = [c for c in plan.conv_log if c.get("winograd") and min(c["k"]) == 1 and not c.get("stride2")]
    assert {c["name"] for c in wino_1d} == ({f"depth.{s}.conv_{a}" for s in ("enc0.1", "enc1.1", "enc2.1", "dec1.1", "dec2.1", "dec4.0") for a in "yx"} |
                                            {"depth.enc0.0.conv_y", "depth.enc0.0.conv_x"})
    # ... and (round 5) the 7-tap stride-2 pair of enc.1.0 as F(4,4) over [even | odd] views (3.5 of 7 multiplies per output and input channel) and the 5 x 1
    # half of enc.2.0 as F(4,3) over [even | odd] rows (3 of 5; its 1 x 5 half stays on the direct kernel: table code 10)
    s2 = [c for c in plan.conv_log if c.get("stride2")]
    assert [c["name"] for c in s2] == ["depth.enc1.0.conv_y", "depth.enc1.0.conv_x", "depth.enc2.0.conv_y"] and all(c["lds"] <= 160 * 1024 for c in s2)
    assert all(c["macs"] * 2 == c["ref_macs"] for c in s2[:2]) and s2[2]["macs"] * 5 == s2[2]["ref_macs"] * 3
    for c in wino_1d:
        m_, r_ = c.get("wino_m", 2), max(c["k"])
        assert (m_, r_) in ((2, 3), (4, 3), (4, 7)) and c["macs"] == c["ref_macs"] * (m_ + r_ - 1) // (m_ * r_) and c["lds"] <= 160 * 1024
        assert c["sig"].startswith(("x", "y")[c["wino_axis"]] + ("_" if r_ == 3 else "7_"))
    assert {max(c["k"]) for c in wino_1d if c.get("wino_m", 2) == 4} == {3, 7}          # both kinds of form are in the measured table
    # ... and the two large layers.Upconv of the mask decoder on the 4-multiply kernel (csrc/conv1d_wino.hip) at 4/16; the other two stay
    # phase-decomposed at 9/16
    wino_u = [c for c in plan.conv_log if c.get("upconv")]
    assert {c["name"] for c in wino_u} == {"mask.dec2.0", "mask.dec3.0"} and all(c["macs"] * 4 == c["ref_macs"] and c["sig"].startswith("u_") for c in wino_u)
    wino_saved = sum(c["ref_macs"] - c["macs"] for c in wino + wino_t + wino_1d + wino_u + s2)
    upconv_phased = sum(c["ref_macs"] for c in plan.conv_log if c["name"] in ("mask.dec0.0", "mask.dec1.0")) / 1e9       # 0.579 of the 3.934 GMAC
    assert abs((plan.conv_macs() + aux + wino_saved) / 1e9 - (61.07 - upconv_phased * 7 / 16)) < 0.01
    direct = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", winograd=False)               # A/B aid: every 3x3 layer on the direct kernel
    assert not any(c.get("winograd") for c in direct.conv_log) and abs((direct.conv_macs() + aux) / 1e9 - (61.07 - 3.934 * 7 / 16)) < 0.01
    assert [n for n, _ in plan.stages["main"]][-1] == "depth.heads" and "apply_mask" not in [n for n, _ in plan.stages["main"]]
    legacy = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", one_channel_kernels=False)       # A/B aid: everything on mr_conv2d_f32
    assert abs(legacy.conv_ref_macs() / 1e9 - 61.07) < 0.01 and not legacy.aux_log and "apply_mask" in [n for n, _ in legacy.stages["main"]]
    assert max(c["lds"] for c in plan.conv_log) <= 160 * 1024
    assert all(c["mb"] in (1, 2, 3, 4, 6) and c["nb"] in (1, 2, 4) and c["split_k"] >= 1 and c["ck"] in (8, 16, 32, 64, 128)
               for c in plan.conv_log if not c.get("winograd"))
    assert sum(1 for c in plan.conv_log if c.get("winograd")) == 9 + 2 + 14 + 2 + 3
    assert sum(c["phases"] == 4 for c in plan.conv_log) == 8        # four Refine transposed convolutions + four phase-decomposed Upconvs
    assert len(plan.stages["encoder"]) + len(plan.stages["encoder_tail"]) == 22 and len(plan.stages["encoder_tail"]) == 5 and plan.stages["cv"][0][0] == "cost_volume" and plan.stages["main"][0][0] == "mask.dec0.0"


def test_stage_launch_lists_fold_convolution_runs(hip_lib):
    """Plan.run_stage walks consecutive convolution-type launches through ONE mr_run_launches call each: at c2 the ~105 C-ABI calls
    of a keyframe become ~25 host calls; every launch is still there, in order, and the items point at the plan's own descriptors."""
    m = MonoRecModel(cv_depth_steps=32)
    plan = engine.Plan(synth.seeded_state_dict(m.state_dict()), 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    calls = launches = 0
    for stage, ops in plan.stages.items():
        steps = plan._compile_stage(stage)
        names = []
        for st in steps:
            calls += 1
            if isinstance(st, tuple):
                items, n, nm = st
                assert n == len(nm) >= 1 and all(items[i].desc for i in range(n)) and all(0 <= items[i].kind <= 6 for i in range(n))
                names += list(nm)
            else:
                names.append(None)
        launches += len(ops)
        assert [n for n in names if n] == [n for n, fn in ops if getattr(fn, "native", None)] and len(names) == len(ops)
    assert launches == 83 and calls <= 30, (launches, calls)
    d = _lib.ConvDesc()
    bad = (_lib.LaunchItem * 2)()
    bad[0].kind, bad[0].desc = 9, ctypes.addressof(d)
    failed = ctypes.c_int32(-1)
    assert hip_lib.mr_run_launches(bad, 1, None, ctypes.byref(failed)) == -1 and failed.value == 0
    bad[0].kind = _lib.LAUNCH_CONV2D                       # an empty descriptor: the entry point's own argument check answers
    assert hip_lib.mr_run_launches(bad, 1, None, ctypes.byref(failed)) == -1 and hip_lib.mr_run_launches(bad, 0, None, None) == 0


def test_bf16_weight_packing_layout(hip_lib):
    """mr_conv_pack_weights_bf16: lane (cout l&15, group g = l>>4), element j = channel 16*c16 + 4*j + g of the chunk, rounded
    to bf16 (nearest even), sources padded to 16 channels (csrc/conv_layout.h)."""
    import numpy as np
    g = torch.Generator().manual_seed(2)
    srcs, cout, kh, kw, mb, ck = [20, 5], 40, 3, 1, 2, 32
    w = torch.randn(cout, sum(srcs), kh, kw, generator=g)
    packed = engine.pack_conv_weight(w, srcs, mb, ck, bf16=True)
    u16 = packed.numpy().view(np.uint16)
    want_bits = (w.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    o = 0
    groups = -(-(-(-cout // 16)) // mb)
    for grp in range(groups):
        cin_off = 0
        for sc in srcs:
            cpad = -(-sc // 16) * 16
            for c0 in range(0, cpad, ck):
                ckq = min(ck, cpad - c0)
                for tap in range(kh * kw):
                    for c16 in range(ckq // 16):
                        for m in range(mb):
                            blk = u16[o:o + 256].reshape(64, 4)
                            o += 256
                            for lane in (0, 17, 35, 63):
                                for j in range(4):
                                    co, cl = (grp * mb + m) * 16 + (lane & 15), c0 + c16 * 16 + 4 * j + (lane >> 4)
                                    exp = want_bits[co, cin_off + cl, tap // kw, tap % kw] if co < cout and cl < sc else 0
                                    assert blk[lane, j] == exp, (grp, sc, c0, tap, c16, m, lane, j)
            cin_off += sc
    assert o == u16.size


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_every_launch_of_a_plan_is_accepted_by_the_library(hip_lib, mode):
    """A plan built on the CPU device packs every weight and sends every conv descriptor through derive() of the library
    (mr_conv2d_lds_bytes) - for the fp32, bf16 and bf16x3 arithmetic, at a small and at the BASELINE configs[1] shape."""
    for (h, w, d) in ((64, 96, 8), (256, 512, 32)):
        m = MonoRecModel(cv_depth_steps=d)
        plan = engine.Plan(synth.seeded_state_dict(m.state_dict()), 1, h, w, 2, d, (0.33, 0.0025), "cpu", bf16=mode)
        assert len(plan.conv_log) == 73 and {int(c["bf16"]) for c in plan.conv_log} == {mode}
        assert all(0 < c["lds"] <= 160 * 1024 for c in plan.conv_log)
        assert abs(plan.conv_macs() - sum(c["macs"] for c in plan.conv_log)) == 0
    if mode == 2:      # no measured table yet: seeded from the bf16 / fp32 entries of the same layer where those still fit
        seeded = sum(1 for c in plan.conv_log if (c["mb"], c["nb"], c["split_k"], c["ck"]) ==
                     tuple(engine.TUNED.get(c["sig"].replace("_bf16x3", "_bf16"), ())[:4]))
        assert seeded >= 40, seeded


def test_bf16x3_weight_packing_layout(hip_lib):
    """mr_conv_pack_weights_bf16x3: the bf16 element order, per lane 4 `hi` = bf16(w) followed by 4 `lo` = bf16(w - hi); hi + lo
    carries 16 mantissa bits of the weight."""
    import numpy as np
    g = torch.Generator().manual_seed(3)
    srcs, cout, kh, kw, mb, ck = [20, 5], 40, 1, 3, 2, 16
    w = torch.randn(cout, sum(srcs), kh, kw, generator=g)
    packed = engine.pack_conv_weight(w, srcs, mb, ck, bf16=2)
    assert packed.numel() == 2 * engine.pack_conv_weight(w, srcs, mb, ck, bf16=1).numel()
    u16 = packed.numpy().view(np.uint16)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    hi_bits, lo_bits = (t.view(torch.int16).numpy().view(np.uint16) for t in (hi, lo))
    assert ((hi.float() + lo.float() - w).abs() <= w.abs() * 2.0 ** -16).all()
    o = 0
    groups = -(-(-(-cout // 16)) // mb)
    for grp in range(groups):
        cin_off = 0
        for sc in srcs:
            cpad = -(-sc // 16) * 16
            for c0 in range(0, cpad, ck):
                ckq = min(ck, cpad - c0)
                for tap in range(kh * kw):
                    for c16 in range(ckq // 16):
                        for m in range(mb):
                            blk = u16[o:o + 512].reshape(64, 8)
                            o += 512
                            for lane in (0, 17, 35, 63):
                                for j in range(4):
                                    co, cl = (grp * mb + m) * 16 + (lane & 15), c0 + c16 * 16 + 4 * j + (lane >> 4)
                                    inside = co < cout and cl < sc
                                    idx = (co, cin_off + cl, tap // kw, tap % kw)
                                    assert blk[lane, j] == (hi_bits[idx] if inside else 0), (grp, sc, c0, tap, c16, m, lane, j)
                                    assert blk[lane, 4 + j] == (lo_bits[idx] if inside else 0), (grp, sc, c0, tap, c16, m, lane, j)
            cin_off += sc
    assert o == u16.size


def test_dropin_rebinds_the_names_the_reference_scripts_look_up(tmp_path):
    """python -m monorec_amd.dropin <script>: a miniature checkout with the reference's import structure (model/model.py re-export,
    model/metric.py star imports, utils/__init__.py star import) - the script itself is untouched."""
    import subprocess
    import sys
    root = tmp_path
    (root / "model" / "monorec").mkdir(parents=True)
    (root / "model" / "metric_functions").mkdir()
    (root / "utils").mkdir()
    (root / "model" / "__init__.py").write_text("")
    (root / "model" / "monorec" / "__init__.py").write_text("")
    (root / "model" / "metric_functions" / "__init__.py").write_text("")
    (root / "model" / "monorec" / "monorec_model.py").write_text("class MonoRecModel:\n    origin = 'reference'\n")
    (root / "model" / "model.py").write_text("from .monorec.monorec_model import MonoRecModel\n")
    (root / "model" / "metric_functions" / "sparse_metrics.py").write_text(
        "def abs_rel_sparse_metric(*a, **k):\n    return 'reference'\ndef other_metric(*a, **k):\n    return 'reference'\n")
    (root / "model" / "metric.py").write_text("from .metric_functions.sparse_metrics import *\n")
    (root / "utils" / "ply_utils.py").write_text("class PLYSaver:\n    origin = 'reference'\n")
    (root / "utils" / "__init__.py").write_text("from .ply_utils import *\n")
    (root / "data_loader").mkdir()
    (root / "data_loader" / "__init__.py").write_text("")
    (root / "data_loader" / "data_loaders.py").write_text("class KittiOdometryDataloader:\n    origin = 'reference'\n")
    (root / "loader_like.py").write_text("import data_loader.data_loaders as module_data\nimport model.model as module_arch\n"
                                         "print(getattr(module_data, 'KittiOdometryDataloader').__module__, module_arch.MonoRecModel.__module__)\n")
    (root / "evaluate_like.py").write_text(
        "import sys\nimport model.metric as module_metric\nimport model.model as module_arch\nfrom utils import PLYSaver\n"
        "cls = getattr(module_arch, 'MonoRecModel')\n"
        "print(cls.__module__, getattr(module_metric, 'abs_rel_sparse_metric').__module__, module_metric.other_metric(),\n"
        "      PLYSaver.__module__, sys.argv[1:])\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "monorec_amd.dropin", "evaluate_like.py", "--config", "x.json"], cwd=root,
                         env=dict(os.environ, PYTHONPATH=repo), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["monorec_amd.model", "monorec_amd.metrics", "reference", "monorec_amd.pointcloud",
                                  "['--config',", "'x.json']"]
    for flags, want in (([], "data_loader.data_loaders"), (["--device-loader"], "monorec_amd.kitti")):    # the loader only on request
        out = subprocess.run([sys.executable, "-m", "monorec_amd.dropin", *flags, "loader_like.py"], cwd=root,
                             env=dict(os.environ, PYTHONPATH=repo), capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split() == [want, "monorec_amd.model"]


def test_python_lds_model_bounds_the_library(hip_lib):
    """engine.candidate_schedules filters by engine.lds_bytes (a Python mirror of derive() in conv_mfma.hip): for every candidate it
    keeps, the library must accept the launch and need no more LDS than the mirror predicted."""
    import itertools
    import random
    rnd = random.Random(5)
    shapes = [(64, [64], 3, 3, 1, 1, 64, 128), (128, [64], 3, 3, 2, 2, 32, 64), (96, [96, 128, 96], 3, 3, 1, 1, 32, 64),
              (48, [35], 7, 1, 1, 1, 256, 512), (64, [48], 1, 7, 1, 2, 128, 256), (1, [24], 3, 3, 1, 1, 256, 512),
              (512, [512], 3, 3, 1, 1, 8, 16), (256, [128], 1, 1, 2, 2, 16, 32), (24, [32], 3, 3, 1, 1, 50, 70)]
    checked = 0
    for (cout, srcs, kh, kw, sh, sw, oh, ow), bf16 in itertools.product(shapes, (0, 1, 2)):      # fp32, bf16, bf16x3
        cands = engine.candidate_schedules(cout, srcs, kh, kw, sh, sw, oh, ow, 1, lds_cap=160 * 1024, bf16=bf16)
        for cd in rnd.sample(cands, min(len(cands), 25)):
            d = _lib.ConvDesc()
            d.num_src = len(srcs)
            for i, c in enumerate(srcs):
                d.src[i], d.src_channels[i] = 16, c
            d.batch, d.src_h, d.src_w = 1, oh * sh, ((ow * sw) + 3) // 4 * 4
            d.kh, d.kw, d.stride_h, d.stride_w, d.pad_top, d.pad_left = kh, kw, sh, sw, kh // 2, kw // 2
            d.out_h, d.out_w, d.dst, d.out_channels, d.dst_total_channels = oh, ow, 16, cout, cout
            d.dst_plane_h, d.dst_plane_w, d.out_step_h, d.out_step_w = oh, ow, 1, 1
            d.packed_weights = 16
            d.cout_blocks_per_wg, d.pixel_blocks_per_wave, d.split_k, d.chunk_channels = cd["mb"], cd["nb"], cd["split_k"], cd["ck"]
            d.waves_per_wg, d.compute_dtype, d.k_split_waves = cd["waves"], bf16, cd.get("kws", 0)
            got = int(hip_lib.mr_conv2d_lds_bytes(ctypes.byref(d)))
            if got == -2 and cd["waves"] == 8:            # 8-wave tiles need the dwordx4 path: legitimately refused for some geometries
                continue
            assert 0 < got <= cd["lds"] <= 160 * 1024, (cout, srcs, kh, kw, cd, got)
            checked += 1
    assert checked > 300


def test_frame_cache_reads_ahead_on_worker_threads():
    """FrameCache(workers=N): decodes run ahead on host threads, every image is decoded once, results equal the serial cache."""
    import threading
    import numpy as np
    from monorec_amd.input_pipeline import FrameCache
    main = threading.get_ident()
    calls, threads = [], set()
    lock = threading.Lock()

    def load(i):
        with lock:
            calls.append(i)
            threads.add(threading.get_ident())
        return np.full((4, 6, 3), i % 251, dtype=np.uint8)

    pre = lambda img: torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)       # stands in for the device preprocessor
    serial = FrameCache(load, pre, capacity=4)
    want = [[float(t[0, 0, 0]) for t in ([serial.sample(i)[0]] + serial.sample(i)[1])] for i in range(1, 30)]
    del calls[:]
    threads.clear()
    cache = FrameCache(load, pre, capacity=4, workers=3, index_range=(0, 31))
    got = []
    for i in range(1, 30):
        kf, fr, idx = cache.sample(i)
        assert idx == [i - 1, i + 1]
        got.append([float(t[0, 0, 0]) for t in [kf] + fr])
    cache.close()
    assert got == want
    assert sorted(calls) == sorted(set(calls)) and set(range(0, 31)) >= set(calls) >= set(range(0, 31 - 1))   # once each, never past the range
    assert cache.decoded == 31 and main not in threads - {main} and len(threads - {main}) >= 1


def test_header_is_a_c_abi_usable_from_plain_c(hip_lib, tmp_path):
    """include/monorec_hip.h compiled as C99 by gcc (-Wall -Wextra -Werror) into a program that links libmonorec_hip.so and drives
    the host-side entry points (weight repack, launch planning, resize tables) - no C++, no torch on that side of the boundary."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(repo, "monorec_amd")
    exe = str(tmp_path / "host_smoke")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(repo, "include"),
                    os.path.join(repo, "tests", "c_abi", "host_smoke.c"), "-o", exe, "-L" + libdir, "-l:libmonorec_hip.so", "-lm",
                    "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("c abi ok"), (out.returncode, out.stdout, out.stderr[-500:])



def test_marching_cost_volume_kernel_codegen():
    """The marching cost-volume kernel lives on two compiler behaviours that a source change can silently lose (it happened while
    trying the reference's conv3d summation order for the box stage): the DPP wave shifts must ride on the adds (v_add_f32_dpp -
    an unfolded v_mov_b32_dpp costs an instruction, a zero-initialised register and the wave-64 occupancy), and the kernel must
    stay at 4 waves per SIMD (<= 128 VGPRs, no scratch).  Checked on the ISA hipcc emits for gfx950."""
    from monorec_amd import build as _build
    flags = dict(_build.SOURCES)["cost_volume.hip"]
    src = os.path.join(ROOT, "monorec_amd", "csrc", "cost_volume.hip")
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([_build._hipcc(), f"--offload-arch={_build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c", src,
                        "-o", os.path.join(d, "cv.o")] + flags, check=True, cwd=d, capture_output=True)
        asm = open(os.path.join(d, "cost_volume-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    # <DP, shared depths, keyframe prepass on / off, exact constant division, relaxed (separable) window sums: the bf16 configuration's variants>
    for variant in ("ILi2ELb0ELb1ELb1ELb0E", "ILi2ELb0ELb0ELb1ELb0E", "ILi1ELb0ELb1ELb1ELb0E", "ILi2ELb0ELb1ELb1ELb1E", "ILi1ELb0ELb1ELb1ELb1E"):
        m = re.search(r"_ZN12_GLOBAL__N_119cv_sad_march_kernel" + variant + r"EEvNS_6CvArgsENS_9MarchGeomE:(.*?)\.Lfunc_end", asm, re.S)
        assert m, variant
        body = m.group(1)
        folded, unfolded = body.count("v_add_f32_dpp"), body.count("v_mov_b32_dpp")
        dp = 2 if variant.startswith("ILi2") else 1
        relaxed = variant.endswith("Lb1E")
        # exact: 6 folded shifts per quantity and step, two steps unrolled; relaxed: 2 per quantity, three steps unrolled
        assert folded >= (55 if relaxed else 100) * dp and unfolded <= 8 * dp, (variant, folded, unfolded)
        meta = asm[asm.index("amdhsa.kernels"):]
        k = re.search(r"cv_sad_march_kernel" + variant + r".*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", meta, re.S)
        assert k and int(k.group(1)) == 0 and int(k.group(2)) <= (128 if dp == 2 else 64), (variant, k and k.groups())   # 4 / 8 waves per SIMD


class _FakePending:
    owned = False          # outputs are views of "resident" tensors: forward() copies them

    def __init__(self, data):
        self._data = data

    def result(self):
        return self._data

    def synchronize(self):
        return self._data


def _stub_submit_one(model, calls):
    """Replace the launch path by a CPU stand-in: outputs are simple functions of the (possibly concatenated) inputs."""
    def submit_one(data, prepared=None, slot=None):
        calls.append(int(data["keyframe"].shape[0]))
        kf = data["keyframe"]
        data["predicted_inverse_depths"] = [kf[:, :1] * 2.0, kf[:, :1, ::2, ::2], kf[:, :1, ::4, ::4], kf[:, :1, ::8, ::8]]
        data["cv_mask"] = kf[:, 1:2] + 1.0
        data["cost_volume"] = kf.repeat(1, 2, 1, 1)
        data["single_frame_cvs"] = [f + 3.0 for f in data["frames"]]
        data["image_features"] = [kf[:, :2]]
        data["inv_depth_min"] = torch.tensor([0.33])
        data["result"], data["mask"] = data["predicted_inverse_depths"][0], data["cv_mask"]
        return _FakePending(data)
    model._submit_one = submit_one


def test_dynamic_batching_groups_requests_and_slices_outputs():
    """Host logic of MonoRecModel(hip_batch_keyframes=K).submit: K equal-shaped requests -> one launch over the concatenated batch,
    every request gets its slice; a result asked for early launches the partial group; a shape change closes the group."""
    m = MonoRecModel(cv_depth_steps=8, hip_batch_keyframes=3).eval()
    calls = []
    _stub_submit_one(m, calls)
    reqs = [dict(keyframe=torch.full((1, 3, 8, 16), float(i)), keyframe_intrinsics=torch.eye(4).unsqueeze(0),
                 keyframe_pose=torch.eye(4).unsqueeze(0), frames=[torch.full((1, 3, 8, 16), 10.0 + i)] * 2,
                 intrinsics=[torch.eye(4).unsqueeze(0)] * 2, poses=[torch.eye(4).unsqueeze(0)] * 2) for i in range(5)]
    hs = [m.submit(r) for r in reqs[:3]]
    assert calls == [3]                                         # the third request filled the group
    for i, h in enumerate(hs):
        out = h.result()
        assert out is reqs[i] and out["result"].shape == (1, 1, 8, 16) and float(out["result"][0, 0, 0, 0]) == 2.0 * i
        assert float(out["single_frame_cvs"][1][0, 0, 0, 0]) == 13.0 + i and out["mask"] is out["cv_mask"]
        assert out["result"] is out["predicted_inverse_depths"][0] and out["inv_depth_min"].shape == (1,)
    h3 = m.submit(reqs[3])
    assert calls == [3] and float(h3.result()["result"][0, 0, 0, 0]) == 6.0 and calls == [3, 1]      # early result -> group of one
    h4 = m.submit(reqs[4])
    other = dict(reqs[0], keyframe=torch.zeros(1, 3, 16, 16), frames=[torch.zeros(1, 3, 16, 16)] * 2)
    h5 = m.submit(other)                                        # another shape: the open group is launched first
    assert calls == [3, 1, 1]
    assert float(h4.result()["result"][0, 0, 0, 0]) == 8.0 and h5.result()["result"].shape == (1, 1, 16, 16) and calls == [3, 1, 1, 1]


def test_dynamic_batching_failed_launch_reaches_every_member_and_bad_requests_fail_at_submit():
    """A coalesced launch that fails must not leave handles without a launch behind: every member's handle re-raises the
    launch's error; a request with a missing key, or a model in training mode, is refused by the submit() that receives it."""
    m = MonoRecModel(cv_depth_steps=8, hip_batch_keyframes=2).eval()

    def boom(data):
        raise ValueError("launch failed")
    m._submit_one = boom
    req = lambda i: dict(keyframe=torch.full((1, 3, 8, 16), float(i)), keyframe_intrinsics=torch.eye(4).unsqueeze(0),
                         keyframe_pose=torch.eye(4).unsqueeze(0), frames=[torch.zeros(1, 3, 8, 16)] * 2,
                         intrinsics=[torch.eye(4).unsqueeze(0)] * 2, poses=[torch.eye(4).unsqueeze(0)] * 2)
    h0 = m.submit(req(0))
    with pytest.raises(ValueError, match="launch failed"):
        m.submit(req(1))                                        # fills the group: the launch fails here ...
    for h in (h0,):
        with pytest.raises(RuntimeError, match="coalesced launch") as e:
            h.result()                                          # ... and every member sees that error, not an AttributeError
        assert isinstance(e.value.__cause__, ValueError)
    assert m._open_group is None
    bad = req(2)
    del bad["poses"]
    with pytest.raises(KeyError):
        m.submit(bad)
    m.train()
    with pytest.raises(NotImplementedError):
        m.submit(req(3))


def test_forward_returns_owned_outputs_with_the_reference_aliasing():
    """forward() on a plan whose outputs cannot be bound to caller-owned memory (handle.owned False: hipGraph replay, the option
    variants with constant buffers) = the enqueued forward + one copy of every output tensor (monorec_model.py:713-727 allocates its
    outputs), with `result is predicted_inverse_depths[0]` and `mask is cv_mask` like the reference (:723-727)."""
    m = MonoRecModel(cv_depth_steps=8)
    _stub_submit_one(m, [])
    m.submit = lambda d: m._submit_one(d)
    m._forward_handle = lambda d: m._submit_one(d)
    import contextlib
    orig = torch.cuda.device
    torch.cuda.device = lambda dev: contextlib.nullcontext()    # no HIP device in this test
    try:
        kf = torch.arange(3 * 8 * 16, dtype=torch.float32).view(1, 3, 8, 16)
        data = dict(keyframe=kf, frames=[kf + 1, kf + 2])
        out = m.forward(data)
    finally:
        torch.cuda.device = orig
    assert out["result"] is out["predicted_inverse_depths"][0] and out["mask"] is out["cv_mask"]
    assert out["result"].data_ptr() != kf.data_ptr() and out["cost_volume"].data_ptr() != kf.data_ptr()
    kf.add_(100.0)                                              # the "resident buffer" changes: owned outputs do not
    assert float(out["result"][0, 0, 0, 1]) == 2.0 and float(out["cv_mask"][0, 0, 0, 0]) == float(8 * 16 + 1)


def test_winograd_weight_packing_and_algebra(hip_lib):
    """mr_wino_pack_weights_f32: U = G g G^T (formed in double, rounded once) in the stream order the kernel reads -
    [cout group][chunk of 8 channels, source-major][position 4a + b][channel quad][cout block][64 lanes], lane = (cout l & 15, channel
    l >> 4) - padded channels / couts zero; and the F(2x2,3x3) identity Y = A^T [(G g G^T) o (B^T d B)] A == correlation, in numpy."""
    import numpy as np
    g = torch.Generator().manual_seed(0)
    cout, srcs = 40, [5, 11]
    w = torch.randn(cout, sum(srcs), 3, 3, generator=g)
    arr = (ctypes.c_int32 * 2)(*srcs)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    U = np.einsum("ai,ocij,bj->ocab", G, w.double().numpy(), G)              # (cout, cin, 4, 4)
    for mbw in (1, 2):
        n = hip_lib.mr_wino_packed_weight_floats(cout, arr, 2, mbw)
        groups, nchunks = -(-cout // (32 * mbw)), 1 + 2                       # 5 -> one chunk of 8, 11 -> two
        assert n == groups * nchunks * 16 * 2 * (2 * mbw) * 64
        packed = torch.full((n,), float("nan"))
        assert hip_lib.mr_wino_pack_weights_f32(w.data_ptr(), cout, arr, 2, mbw, packed.data_ptr()) == 0
        P = packed.view(groups, nchunks, 16, 2, 2 * mbw, 64).numpy()
        assert np.isfinite(P).all()
        want = np.zeros_like(P)
        for co in range(cout):
            for ci in range(sum(srcs)):
                chunk, cl = (0, ci) if ci < 5 else (1 + (ci - 5) // 8, (ci - 5) % 8)
                gi, mb, l15 = co // (32 * mbw), (co % (32 * mbw)) // 16, co % 16
                want[gi, chunk, :, cl // 4, mb, (cl % 4) * 16 + l15] = U[co, ci].reshape(16)
        assert np.abs(P - want).max() <= 1e-7                                 # one rounding to fp32; every padded slot exactly zero
        assert (P[want == 0] == 0).all()
    assert hip_lib.mr_wino_packed_weight_floats(cout, arr, 2, 3) == 0 and hip_lib.mr_wino_pack_weights_f32(w.data_ptr(), cout, arr, 2, 4, packed.data_ptr()) == -1
    rng = np.random.default_rng(0)
    d, gk = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]])
    Y = At @ ((G @ gk @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[(d[i:i + 3, j:j + 3] * gk).sum() for j in range(2)] for i in range(2)])
    assert np.abs(Y - ref).max() < 1e-12


def test_c3_plan_runs_the_large_3x3_layers_as_f44(hip_lib):
    """The c3 shape (batch 8, 4 frames, 64 bins): the table measured in tools/sessions/r04_s18.sh sends every full- and half-resolution 3x3
    stride-1 layer to F(4x4,3x3) (csrc/conv_wino44.hip: 36 of 144 multiplies per 4x4 outputs, one 153 KB workgroup per CU)."""
    m = MonoRecModel(cv_depth_steps=64)
    plan = engine.Plan(synth.seeded_state_dict(m.state_dict()), 8, 256, 512, 4, 64, (0.33, 0.0025), "cpu")
    f44 = [c for c in plan.conv_log if c.get("wino_variant") == 3]
    assert {c["name"] for c in f44} == {"mask.enc0.0", "mask.enc0.1", "mask.enc1.0", "mask.enc1.1", "mask.enc2.0", "mask.enc2.1", "mask.dec1.1",
                                        "mask.dec1.2", "mask.dec2.1", "mask.dec2.2", "mask.dec3.1", "mask.dec3.2", "depth.dec4.2"}
    assert all(c["macs"] * 4 == c["ref_macs"] and c["lds"] <= 160 * 1024 and c["wgs"] >= 192 for c in f44)


def test_polyphase_stride2_forms_are_exact():
    """monorec_amd.cooktoom: the odd tile sizes (F(2,2), F(2,4), F(4,2), F(4,4)) satisfy the bilinear identity exactly with dyadic B^T / A^T, and a
    stride-2 correlation equals the sum of the two phase filters evaluated through them (exact rational arithmetic; pads of both parities)."""
    import random
    from fractions import Fraction
    from monorec_amd import cooktoom as ct
    for m, r in ((2, 2), (2, 4), (4, 2), (4, 4)):
        at, g, bt = ct.cook_toom(m, r)
        assert ct.identity_holds(m, r, at, g, bt)
        assert all((Fraction(v).denominator & (Fraction(v).denominator - 1)) == 0 for mat in (at, bt) for row in mat for v in row)
    rnd = random.Random(3)
    for r, pad in ((7, 2), (7, 3), (5, 1), (5, 2), (3, 0), (3, 1)):
        for m in (2, 4):
            for n in (23, 24):
                d = [Fraction(rnd.randint(-9, 9)) for _ in range(n)]
                g = [Fraction(rnd.randint(-5, 5), rnd.randint(1, 4)) for _ in range(r)]
                ref = [sum(g[k] * (d[2 * i + k - pad] if 0 <= 2 * i + k - pad < n else 0) for k in range(r)) for i in range(-(-n // 2))]
                assert ct.correlate_stride2_polyphase(d, g, m, pad) == ref, (r, pad, m, n)
    assert ct.polyphase_stride2(7, 2) == ([0, 2, 4, 6], [1, 3, 5]) and ct.polyphase_stride2(5, 1) == ([0, 2, 4], [1, 3])


def test_winograd_choice_table_and_rule():
    """engine.choose_winograd: the measured table wins (c2: the two full-resolution mask stages and the big decoder layers go to the
    Winograd kernel, every ResNet layer of a batch-1 keyframe stays on the direct kernel); unknown shapes follow the workgroup-count
    rule; widths that are not a multiple of 4 never qualify."""
    assert engine.WINOGRAD, "monorec_amd/tuned_winograd.json missing"
    # 3x3 / transposed keys: + 10 = input transform in registers, + 20 = ... with 16-channel tail workgroups; 1-D keys: 10 m + blocks = F(m, taps)
    assert set(engine.WINOGRAD.values()) <= {0, 1, 2, 3, 4, 10, 11, 12, 14, 21, 22, 23, 24, 31, 41, 42, 43, 44}      # 31: F(4x4,3x3); 10 (s2k keys): only the k x 1 half
    assert all(k.startswith("s2k") for k, v in engine.WINOGRAD.items() if v == 10)
    assert all(v in (0, 21, 22, 23, 24, 41, 42, 43) for k, v in engine.WINOGRAD.items() if k[:3] in ("x7_", "y7_"))
    assert engine.choose_winograd_1d(0, 48, [48], 256, 512, 1) in (3, 41, 42, 43) and engine.choose_winograd_1d(1, 256, [256], 16, 32, 1) == 0 and engine.choose_winograd_1d(0, 48, [48], 256, 510, 1) == 0
    assert engine.choose_winograd_1d(0, 48, [48], 256, 512, 1, 7) == 43 and engine.choose_winograd_1d(1, 48, [32, 3], 256, 512, 1, 7) == 43      # depth.enc0.0 @ c2: F(4,7)
    assert engine.choose_winograd_1d(0, 48, [48], 128, 256, 1, 7) == 43                                                                           # unknown 7-tap shape: the nearest measured one's form (round 6)
    assert engine.choose_winograd_t(48, [64, 64, 64], 128, 256, 1) % 10 in (1, 2) and engine.choose_winograd_t(256, [256], 16, 32, 1) == 0   # Refine: depth.dec3 / dec0 @ c2
    assert engine.choose_winograd_t(48, [64, 64, 64], 100, 256, 1) == 21 and engine.choose_winograd_t(48, [64], 128, 254, 1) == 0            # unknown shape: nearest signature's form / width % 4: direct
    assert engine.choose_winograd(32, [32], 256, 512, 2) == 31 and engine.choose_winograd(48, [32, 64], 256, 512, 1) == 31  # mask.enc0.*, mask.dec3.1 @ c2: F(4x4,3x3)
    assert engine.choose_winograd(64, [64], 64, 128, 1) == 0 and engine.choose_winograd(512, [512], 8, 16, 1) == 0          # ResNet l1 / l4 @ c2
    assert engine.choose_winograd(64, [64], 256, 512, 32) == 31                                                              # mask.enc0.* @ c3: F(4x4,3x3) (r04_s18)
    # shapes without an entry take the form measured for the NEAREST signature (round 6); only when nothing is within reach the old
    # workgroup-count rule decides
    assert engine.choose_winograd(32, [32], 64, 96, 1) == 0            # nearest: a small 32-channel layer that stayed on the direct kernel
    assert engine.choose_winograd(32, [32], 256, 768, 3) == 41 and engine.choose_winograd(96, [96], 256, 768, 3) == 31      # nearest: full-resolution layers on F(4x4,3x3)
    assert engine.choose_winograd(7, [5], 4000, 4000, 64) == 11        # nothing within reach: 2 M workgroups, <= 32 couts -> transform in registers
    assert engine.choose_winograd(32, [32], 256, 510, 4) == 0          # width % 4


def test_rule_path_agrees_with_the_tables_it_is_derived_from():
    """Leave-one-out replay of every table key through the nearest-signature rules: with a key hidden, the rule must still send the layer
    to the same kernel FAMILY (direct vs reduced-multiply form) for most keys and pick the same register tile / wave count for a third of the direct
    kernel's schedules (the tables hold many near-ties, so exact agreement is not the measure - tools/sessions/r06_* measure the time)."""
    import collections
    same, fam, tot = collections.Counter(), collections.Counter(), collections.Counter()
    for key, code in engine.WINOGRAD.items():
        m = engine._WSIG_RE.match(key)
        assert m, key
        pre, cis = m.group(1) or "", [int(c) for c in m.group(3).split("+")]
        pred = engine.nearest_form(pre, int(m.group(2)), sum(cis), int(m.group(4)) * int(m.group(5)), int(m.group(6)), exclude=(key,))
        tot[pre] += 1
        same[pre] += pred == code
        fam[pre] += pred is not None and (pred == 0) == (code == 0)
    assert sum(tot.values()) == len(engine.WINOGRAD)
    assert same[""] >= 0.55 * tot[""] and fam[""] >= 0.8 * tot[""], (same[""], fam[""], tot[""])
    assert sum(fam.values()) >= 0.8 * sum(tot.values()), (fam, tot)
    pad = lambda t: (tuple(t) + (4, 0, 0))[:7] if len(t) < 5 else (tuple(t) + (0, 0))[:7]
    n = tile = 0
    for key, sched in engine.TUNED.items():
        m = engine._SIG_RE.match(key)
        assert m, key
        cis = [int(c) for c in m.group(2).split("+")]
        mode = {"": 0, "_bf16": 1, "_bf16x3": 2}[m.group(12) or ""]
        c = engine.nearest_schedules(int(m.group(1)), cis, int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8)),
                                     int(m.group(9)), int(m.group(10)), mode, m.group(11) == "u", exclude=(key,), limit=1)
        n += 1
        tile += bool(c) and pad(c[0])[:2] == pad(sched)[:2] and pad(c[0])[4] == pad(sched)[4]
    assert n == len(engine.TUNED) and tile >= 0.3 * n, (tile, n)


def test_f2_table_documents_its_coverage():
    """`hip_exact_convs="f2"`: tuned_winograd_f2.json holds MEASURED F(2,.) choices only for the c2 / c3 keys whose
    main-table entry is a larger form; every other such key falls back to a rule (variant 11 for a 3x3 layer, the direct kernel for a 1-D
    layer) whose throughput is unmeasured - the documentation says so with these counts, and the fallback itself is pinned here."""
    # (the `s2k` keys - stride-2 pairs, round 5 - are not part of this: under "f2" / "direct" those layers simply stay on the direct kernel)
    larger = [k for k, v in engine.WINOGRAD.items() if (v >= 40 or v // 10 == 3) and not k.startswith("s2k")]
    covered = [k for k in larger if k in engine.WINOGRAD_F2]
    assert set(engine.WINOGRAD_F2) <= set(larger)
    assert all(k.endswith("_b1") or k.endswith("_b8") for k in covered)                  # the c2 / c3 keys
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert f"{len(covered)} of the {len(larger)} keys" in doc, (len(covered), len(larger))
    uncovered3 = next(k for k in larger if k not in engine.WINOGRAD_F2 and k.startswith("co"))
    m = re.match(r"co(\d+)_ci([\d+]+)_o(\d+)x(\d+)_b(\d+)", uncovered3)
    cout, srcs, h, w, b = int(m.group(1)), [int(c) for c in m.group(2).split("+")], int(m.group(3)), int(m.group(4)), int(m.group(5))
    assert engine.choose_winograd(cout, srcs, h, w, b, f2=True) == 11
    uncovered1 = next(k for k in larger if k not in engine.WINOGRAD_F2 and k[:2] in ("x_", "y_"))
    m = re.match(r"([xy])_co(\d+)_ci([\d+]+)_o(\d+)x(\d+)_b(\d+)", uncovered1)
    assert engine.choose_winograd_1d("xy".index(m.group(1)), int(m.group(2)), [int(c) for c in m.group(3).split("+")], int(m.group(4)), int(m.group(5)),
                                     int(m.group(6)), 3, f2=True) == 0


def test_winograd_1d_weight_packing_and_algebra(hip_lib):
    """mr_wino1d_pack_weights_f32: U = G g (double, rounded once) in the stream order conv1d_wino.hip reads - [cout group of 16 mbw]
    [chunk of 8 channels, source-major][position][channel quad][cout block][64 lanes], lane = (cout l & 15, channel l >> 4) - and the
    F(2,3) identity y = A^T [(G g) o (B^T d)] == correlation, evaluated from the packed stream in numpy along both axes."""
    g = torch.Generator().manual_seed(11)
    srcs_c, cout, mbw = [5, 11], 40, 2
    cin = sum(srcs_c)
    for axis, kk in ((0, (1, 3)), (1, (3, 1))):
        w = torch.randn(cout, cin, *kk, generator=g)
        sc = (ctypes.c_int32 * len(srcs_c))(*srcs_c)
        n = hip_lib.mr_wino1d_packed_weight_floats(cout, sc, len(srcs_c), mbw)
        cpads = [(c + 7) // 8 * 8 for c in srcs_c]
        groups = (cout + 16 * mbw - 1) // (16 * mbw)
        assert n == groups * sum(cpads) // 8 * (4 * 2 * mbw * 64)
        packed = torch.empty(n)
        _lib.check(hip_lib.mr_wino1d_pack_weights_f32(w.data_ptr(), cout, sc, len(srcs_c), mbw, packed.data_ptr()))
        U = np.zeros((4, groups * mbw * 16, sum(cpads)), np.float64)          # [p][cout][padded cin]
        st = packed.numpy().reshape(groups, sum(cpads) // 8, 4, 2, mbw, 64)
        for gi in range(groups):
            for q in range(sum(cpads) // 8):
                for p_ in range(4):
                    for c4 in range(2):
                        for m in range(mbw):
                            for lane in range(64):
                                U[p_, (gi * mbw + m) * 16 + (lane & 15), q * 8 + c4 * 4 + (lane >> 4)] = st[gi, q, p_, c4, m, lane]
        x = torch.randn(1, cin, 6, 8, generator=g)
        ref = F.conv2d(x, w, padding=(kk[0] // 2, kk[1] // 2)).numpy()[0]
        xp = np.zeros((sum(cpads), 6 + 2, 8 + 2))                             # padded channels (zero) + halo
        off = 0
        for s_, (c, cp) in enumerate(zip(srcs_c, cpads)):
            xp[off:off + c, 1:-1, 1:-1] = x[0, sum(srcs_c[:s_]):sum(srcs_c[:s_]) + c].numpy()
            off += cp
        out = np.zeros((cout, 6, 8))
        Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
        for y in range(0, 6, 2 if axis == 1 else 1):
            for xx in range(0, 8, 2 if axis == 0 else 1):
                d = xp[:, y + 1, xx:xx + 4] if axis == 0 else xp[:, y:y + 4, xx + 1]       # inputs j - 1 .. j + 2 along the axis
                v = d @ Bt.T                                                                # (cin, 4)
                mprod = np.einsum("pok,kp->po", U, v)                                       # (4, cout)
                yy = At @ mprod                                                             # (2, cout)
                if axis == 0:
                    out[:, y, xx:xx + 2] = yy[:, :cout].T
                else:
                    out[:, y:y + 2, xx] = yy[:, :cout].T
        assert np.abs(out - ref).max() <= 1e-5, axis


def test_cooktoom_header_is_what_its_generator_writes_and_the_forms_are_exact(tmp_path):
    """csrc/cooktoom_1d.h (transform chains + G tables of F(4,3), F(2,7), F(4,7), F(4,4)) == monorec_amd.cooktoom.generate_header(); the forms
    satisfy the bilinear identity in exact rational arithmetic with dyadic A^T / B^T; and the GENERATED code itself, compiled for the host
    (tests/c_abi/cooktoom_host.cpp), reproduces the r-tap correlation in fp32 to the rounding the numerics study expects."""
    from fractions import Fraction
    from monorec_amd import cooktoom
    with open(os.path.join(ROOT, "monorec_amd", "csrc", "cooktoom_1d.h")) as f:
        assert f.read().rstrip("\n") == cooktoom.generate_header().rstrip("\n"), "run python tools/gen_cooktoom.py"
    for m, r in cooktoom.FORMS + ((2, 3),):
        at, g, bt = cooktoom.cook_toom(m, r)
        assert cooktoom.identity_holds(m, r, at, g, bt)
        assert all(v.denominator & (v.denominator - 1) == 0 for row in at + bt for v in row)          # exact fp32 literals
    at, g, bt = cooktoom.cook_toom(2, 3)                        # F(2,3) up to the scaling conv1d_wino.hip writes out (G rows x -1 / B^T rows x -1)
    assert [[abs(v) for v in row] for row in g] == [[1, 0, 0], [Fraction(1, 2)] * 3, [Fraction(1, 2)] * 3, [0, 0, 1]]
    exe = str(tmp_path / "cooktoom_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "c_abi", "cooktoom_host.cpp")], check=True)
    rows = [ln.split() for ln in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()]
    assert [(int(a), int(b)) for a, b, _, _ in rows] == list(cooktoom.FORMS)
    bars = {(4, 3): 5e-6, (2, 7): 1e-5, (4, 7): 2e-4, (4, 4): 5e-6}     # single products of O(1) values; sums over channels average it down
    for a, b, err, scale in rows:
        assert float(err) <= bars[(int(a), int(b))] and float(scale) > 1.0, (a, b, err)


def test_cooktoom_weight_packing_and_plan_routing(hip_lib, monkeypatch):
    """mr_cooktoom1d_pack_weights_f32: U = G g (double, rounded once) in the stream order of mr_wino1d_pack_weights_f32 with m + r - 1
    positions; and the plan sends the 7-tap layers of DepthModule.enc.0.0 (monorec_model.py:487-500) to the form the table names, with the
    executed multiply-adds (m + r - 1) / (m r) of the reference's and an LDS request the library accepts."""
    from monorec_amd import cooktoom
    g = torch.Generator().manual_seed(12)
    srcs_c, cout, mbw = [5, 11], 40, 2
    cin = sum(srcs_c)
    sc = (ctypes.c_int32 * len(srcs_c))(*srcs_c)
    cpads = [(c + 7) // 8 * 8 for c in srcs_c]
    groups = (cout + 16 * mbw - 1) // (16 * mbw)
    for m, r in cooktoom.FORMS:
        npos = m + r - 1
        G = np.array([[float(v) for v in row] for row in cooktoom.cook_toom(m, r)[1]])
        w = torch.randn(cout, cin, 1, r, generator=g)
        n = hip_lib.mr_cooktoom1d_packed_weight_floats(cout, sc, len(srcs_c), mbw, m, r)
        if (m, r) == (2, 7) and not hip_lib.has_diagnostic_forms:      # F(2,7): diagnostic library only since round 4 (no table ever selected it)
            assert n == 0
            continue
        assert n == groups * sum(cpads) // 8 * (npos * 2 * mbw * 64)
        packed = torch.empty(n)
        _lib.check(hip_lib.mr_cooktoom1d_pack_weights_f32(w.data_ptr(), cout, sc, len(srcs_c), mbw, m, r, packed.data_ptr()))
        st = packed.numpy().reshape(groups, sum(cpads) // 8, npos, 2, mbw, 64)
        want = np.einsum("pj,ocj->poc", G, w[:, :, 0, :].double().numpy())          # [p][cout][cin]
        off, cin_off = 0, 0
        for c, cp in zip(srcs_c, cpads):
            for cl in range(cp):
                q, c4, hi = (off + cl) // 8, ((off + cl) % 8) // 4, (off + cl) % 4
                for co in range(groups * mbw * 16):
                    got = st[co // (16 * mbw), q, :, c4, (co // 16) % mbw, hi * 16 + co % 16]
                    exp = want[:, co, cin_off + cl] if (co < cout and cl < c) else np.zeros(npos)
                    assert np.allclose(got, exp, rtol=2e-7, atol=1e-9), (m, r, co, cl)      # one fp32 ulp: order of the double sum over the taps
            off, cin_off = off + cp, cin_off + c
    # F(4,4) has 7 positions: with an odd number of blocks per wave a chunk's U block is padded with zeros to whole 1 KiB DMA pieces
    n1 = hip_lib.mr_cooktoom1d_packed_weight_floats(cout, sc, len(srcs_c), 1, 4, 4)
    assert n1 == ((cout + 15) // 16) * (sum(cpads) // 8) * 1024
    w = torch.randn(cout, cin, 1, 4, generator=g)
    packed = torch.full((n1,), float("nan"))
    _lib.check(hip_lib.mr_cooktoom1d_pack_weights_f32(w.data_ptr(), cout, sc, len(srcs_c), 1, 4, 4, packed.data_ptr()))
    st = packed.numpy().reshape(-1, 1024)
    assert not np.isnan(st).any() and (st[:, 7 * 2 * 64:] == 0).all() and (st[:, :7 * 2 * 64] != 0).any()
    assert hip_lib.mr_cooktoom1d_packed_weight_floats(cout, sc, len(srcs_c), mbw, 2, 3) == 0                # F(2,3): the other entry point
    assert hip_lib.mr_cooktoom1d_packed_weight_floats(cout, sc, len(srcs_c), 4, 4, 7) == 0                  # F(4,7): at most 3 blocks per wave
    mdl = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(mdl.state_dict())
    for key in [k for k in engine.WINOGRAD if k[:3] in ("x7_", "y7_")]:
        monkeypatch.delitem(engine.WINOGRAD, key)                  # a table without 7-tap entries: those layers stay on the direct kernel
    base = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    seven = [c for c in base.conv_log if max(c["k"]) == 7 and tuple(c["spec"]["stride"]) == (1, 1)]
    assert [c["name"] for c in seven] == ["depth.enc0.0.conv_y", "depth.enc0.0.conv_x"] and not any(c.get("winograd") for c in seven)
    codes = (43, 22) if hip_lib.has_diagnostic_forms else (43, 42)       # F(2,7) (codes 2x on a 7-tap key): diagnostic library only
    for c, code in zip(seven, codes):
        axis = 0 if c["k"][0] == 1 else 1
        monkeypatch.setitem(engine.WINOGRAD, ("x7_", "y7_")[axis] + engine.winograd_signature(c["cout"], [s_[1] for s_ in c["spec"]["src_shapes"]], 256, 512, 1), code)
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    routed = {c["name"]: c for c in plan.conv_log if max(c["k"]) == 7 and c.get("winograd") and not c.get("stride2")}     # (stride-2 enc.1.0: its own test)
    assert set(routed) == {"depth.enc0.0.conv_y", "depth.enc0.0.conv_x"}
    cy, cx = routed["depth.enc0.0.conv_y"], routed["depth.enc0.0.conv_x"]
    assert cy["wino_m"] == 4 and cy["macs"] * 28 == cy["ref_macs"] * 10 and cy["sig"].startswith("y7_") and 0 < cy["lds"] <= 160 * 1024
    if codes[1] == 22:
        assert cx["wino_m"] == 2 and cx["macs"] * 14 == cx["ref_macs"] * 8 and cx["sig"].startswith("x7_") and 0 < cx["lds"] <= 160 * 1024
    else:
        assert cx["wino_m"] == 4 and cx["macs"] * 28 == cx["ref_macs"] * 10 and cx["sig"].startswith("x7_") and 0 < cx["lds"] <= 160 * 1024
    assert abs(plan.conv_ref_macs() - base.conv_ref_macs()) == 0


def test_stride2_layers_as_stride1_forms_over_even_odd_views(hip_lib, monkeypatch):
    """Round 5: a stride-2 ConvReLU2 pair (reference model/layers.py:289-314, monorec_model.py:489-501) on the stride-1 Cook-Toom kernel.
    (1) engine.stride2_unified_weights + the view / split conventions of Plan._conv_relu2_stride2, replayed with plain torch slicing: even / odd
        rows concatenated on channels, a ceil(k/2)-tap 'same'-style filter with 1 zero in front, the result split by column parity, the same
        again along x - equals the strided pair exactly up to fp32 summation order;
    (2) the plan routes the pair when (and only when) the table names its shape, keeps the reference's multiply-adds on the books, executes
        (3 + ceil(k/2)) / 4 multiplies per output and channel PAIR, and the library accepts both launches (strided views, column-split mid)."""
    g = torch.Generator().manual_seed(21)
    for k, (h, w), (cin, cmid, cout) in ((7, (12, 16), (5, 6, 4)), (5, (8, 24), (3, 7, 5))):
        x = torch.randn(2, cin, h, w, generator=g, dtype=torch.float64)
        wy, wx = torch.randn(cmid, cin, k, 1, generator=g, dtype=torch.float64), torch.randn(cout, cmid, 1, k, generator=g, dtype=torch.float64)
        pt, pb = engine.same_pad(h, k, 2)
        pl, pr = engine.same_pad(w, k, 2)
        t = F.conv2d(F.pad(x, (0, 0, pt, pb)), wy, stride=(2, 1))
        ref = F.conv2d(F.pad(t, (pl, pr, 0, 0)), wx, stride=(1, 2))
        r2 = (k + 1) // 2
        uy = engine.stride2_unified_weights(wy.float(), h, axis=1).double()
        ux = engine.stride2_unified_weights(wx.float(), w, axis=0).double()
        assert tuple(uy.shape) == (cmid, 2 * cin, r2, 1) and tuple(ux.shape) == (cout, 2 * cmid, 1, r2)
        rows = torch.cat([x[:, :, 0::2], x[:, :, 1::2]], 1)                                  # [even rows | odd rows]: views, no copy, in the kernel
        t2 = F.conv2d(F.pad(rows, (0, 0, 1, r2 - 2)), uy.double())
        assert float((t2 - t).abs().max()) < 1e-5
        cols = torch.cat([t2[..., 0::2], t2[..., 1::2]], 1)                                  # the column-split intermediate
        got = F.conv2d(F.pad(cols, (1, r2 - 2, 0, 0)), ux.double())
        assert tuple(got.shape) == tuple(ref.shape) and float((got - ref).abs().max()) < 1e-5, (k, float((got - ref).abs().max()))
    mdl = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(mdl.state_dict())
    for key in [k_ for k_ in engine.WINOGRAD if k_.startswith("s2k")]:
        monkeypatch.delitem(engine.WINOGRAD, key)
    base = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    s2 = [c for c in base.conv_log if tuple(c["spec"]["stride"]) != (1, 1) and c["name"].startswith("depth.enc")]
    assert [c["name"] for c in s2] == [f"depth.enc{i}.0.conv_{a}" for i in (1, 2, 3, 4) for a in "yx"] and not any(c.get("winograd") for c in s2)
    monkeypatch.setitem(engine.WINOGRAD, engine.stride2_signature(7, 64, 48, 128, 256, 1), 24)       # depth.enc1.0
    monkeypatch.setitem(engine.WINOGRAD, engine.stride2_signature(5, 128, 64, 64, 128, 1), 41)       # depth.enc2.0
    monkeypatch.setitem(engine.WINOGRAD, engine.stride2_signature(5, 192, 128, 32, 64, 1), 0)        # depth.enc3.0: measured, stays direct (no entry = the nearest entry's form)
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu")
    routed = {c["name"]: c for c in plan.conv_log if c.get("stride2")}
    assert set(routed) == {"depth.enc1.0.conv_y", "depth.enc1.0.conv_x", "depth.enc2.0.conv_y", "depth.enc2.0.conv_x"}
    cy, cx = routed["depth.enc1.0.conv_y"], routed["depth.enc1.0.conv_x"]
    assert (cy["mb"], cx["mb"], cy["wino_m"], cy["wino_taps"], tuple(cy["k"]), tuple(cx["k"])) == (2, 4, 4, 4, (7, 1), (1, 7))
    assert cy["macs"] * 2 == cy["ref_macs"] and cx["macs"] * 2 == cx["ref_macs"]                   # F(4,4) over channel pairs: 3.5 of 7 multiplies
    c5 = routed["depth.enc2.0.conv_y"]
    assert c5["wino_taps"] == 3 and c5["macs"] * 5 == c5["ref_macs"] * 3 and 0 < c5["lds"] <= 160 * 1024      # F(4,3): 3 of 5
    assert plan.conv_ref_macs() == base.conv_ref_macs() and plan.conv_macs() < base.conv_macs()
    assert plan.launch_stamp() != base.launch_stamp()
    # hip_exact_convs modes never take the larger forms
    for forms in ("f2", "direct"):
        p2 = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", conv_forms=forms)
        assert not any(c.get("stride2") for c in p2.conv_log)


def test_winograd44_weight_packing(hip_lib):
    """mr_wino44_pack_weights_f32: U = G g G^T (6 x 6, G of F(4,3): cooktoom.py) in double, rounded once, in the stream order conv_wino44.hip
    reads - [group of 32 couts][chunk of 8 channels, source-major][channel quad][block of 16][j][64 lanes][i], position p = 6 i + j."""
    from monorec_amd import cooktoom
    G = np.array([[float(v) for v in row] for row in cooktoom.cook_toom(4, 3)[1]])
    g = torch.Generator().manual_seed(13)
    srcs_c, cout = [5, 11], 40
    cin = sum(srcs_c)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    sc = (ctypes.c_int32 * len(srcs_c))(*srcs_c)
    cpads = [(c + 7) // 8 * 8 for c in srcs_c]
    groups = (cout + 31) // 32
    n = hip_lib.mr_wino44_packed_weight_floats(cout, sc, len(srcs_c))
    assert n == groups * sum(cpads) // 8 * (36 * 2 * 2 * 64)
    packed = torch.empty(n)
    _lib.check(hip_lib.mr_wino44_pack_weights_f32(w.data_ptr(), cout, sc, len(srcs_c), packed.data_ptr()))
    st = packed.numpy().reshape(groups, sum(cpads) // 8, 2, 2, 6, 64, 6)      # [group][chunk][quad][block][j][lane][i], position p = 6 i + j
    want = np.einsum("ia,ocab,jb->ijoc", G, w.double().numpy(), G).reshape(36, cout, cin)
    off, cin_off = 0, 0
    for c, cp in zip(srcs_c, cpads):
        for cl in range(cp):
            q, c4, hi = (off + cl) // 8, ((off + cl) % 8) // 4, (off + cl) % 4
            for co in range(groups * 32):
                got = st[co // 32, q, c4, (co // 16) % 2, :, hi * 16 + co % 16, :].T.reshape(36)     # (j, i) -> p = 6 i + j
                exp = want[:, co, cin_off + cl] if (co < cout and cl < c) else np.zeros(36)
                assert np.allclose(got, exp, rtol=3e-7, atol=1e-9), (co, cl)
        off, cin_off = off + cp, cin_off + c
    assert hip_lib.mr_wino44_packed_weight_floats(0, sc, len(srcs_c)) == 0
    d = _lib.WinoDesc()
    assert hip_lib.mr_conv3x3_winograd44_lds_bytes(ctypes.byref(d)) == -1            # empty descriptor: bad argument, nothing launched


def _check_host_geometry_against_the_reference_form():
    from monorec_amd.model import host_geometry_reference_form
    import monorec_amd.model as mm
    for hard in (False, True):
        for b, f, seed in ((1, 2, 3), (2, 4, 4), (8, 4, 5), (1, 1, 6)):
            batch = synth.make_batch(b, 64, 96, f, seed=seed, hard_pose=hard)
            a = (batch["keyframe_intrinsics"], batch["keyframe_pose"], batch["intrinsics"], batch["poses"])
            mm._KINV_CACHE.clear()
            k0, p0 = host_geometry_reference_form(*a)
            for _ in range(2):                               # second call: the intrinsics inverse comes from the cache
                k1, p1 = host_geometry(*a)
                assert torch.equal(k1, k0) and torch.equal(p1, p0), (hard, b, f)
    g = torch.Generator().manual_seed(0)
    for trial in range(200):                                 # random rigid poses up to ~100 m from the origin, one batched inversion vs F calls
        poses = []
        for _ in range(3):
            m = torch.eye(4)
            m[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
            m[:3, 3] = torch.randn(3, generator=g) * (100.0 if trial % 2 else 1.0)
            poses.append(m.unsqueeze(0))
        k = torch.tensor([[[489.23, 0, 248.31, 0], [0, 489.23, 126.69, 0], [0, 0, 1, 0], [0, 0, 0, 1]]])
        a = (k, poses[0], [k, k], poses[1:])
        assert all(torch.equal(x, y) for x, y in zip(host_geometry(*a), host_geometry_reference_form(*a))), trial


def test_winograd44s_weight_packing_holds_the_same_transformed_filters(hip_lib):
    """mr_wino44s_pack_weights_f32 (csrc/conv_wino44s.hip: F(4x4,3x3) with the positions of a tile split over two waves): the same U = G g G^T values
    as mr_wino44_pack_weights_f32, re-ordered [group of 32 couts][chunk of 4 channels, source-major][block of 16][position half][row ii of the
    half][column half jh][64 lanes][4] with position p = 6 (3 half + ii) + 4 jh + e (zero pads for 4 jh + e >= 6)."""
    from monorec_amd import cooktoom
    g = torch.Generator().manual_seed(14)
    srcs_c, cout = [6, 9], 40
    cin = sum(srcs_c)
    sc = (ctypes.c_int32 * len(srcs_c))(*srcs_c)
    cp4 = [(c + 3) // 4 * 4 for c in srcs_c]
    groups = (cout + 31) // 32
    w = torch.randn(cout, cin, 3, 3, generator=g)
    n = hip_lib.mr_wino44s_packed_weight_floats(cout, sc, len(srcs_c))
    assert n == groups * (sum(cp4) // 4) * 6144
    packed = torch.full((n,), float("nan"))
    _lib.check(hip_lib.mr_wino44s_pack_weights_f32(w.data_ptr(), cout, sc, len(srcs_c), packed.data_ptr()))
    st = packed.numpy().reshape(groups, sum(cp4) // 4, 2, 2, 3, 2, 64, 4)        # [g][chunk][block][half][ii][jh][lane][e]
    G = np.array([[float(v) for v in row] for row in cooktoom.cook_toom(4, 3)[1]])
    want = np.einsum("pi,ocij,qj->pqoc", G, w.double().numpy(), G)               # U[pi][pj][cout][cin]
    assert not np.isnan(st).any()
    off, cin_off = 0, 0
    for c, cp in zip(srcs_c, cp4):
        for cl in range(cp):
            q, hi = (off + cl) // 4, (off + cl) % 4
            for co in range(groups * 32):
                lane = hi * 16 + co % 16
                for hf in range(2):
                    for ii in range(3):
                        for jh in range(2):
                            for e in range(4):
                                pj = 4 * jh + e
                                got = st[co // 32, q, (co // 16) % 2, hf, ii, jh, lane, e]
                                exp = want[3 * hf + ii, pj, co, cin_off + cl] if (pj < 6 and co < cout and cl < c) else 0.0
                                assert abs(got - exp) <= 2e-7 * max(1.0, abs(exp)), (co, cl, hf, ii, pj)
        off, cin_off = off + cp, cin_off + c
    assert hip_lib.mr_wino44s_packed_weight_floats(0, sc, len(srcs_c)) == 0
    d = _lib.WinoDesc()
    assert hip_lib.mr_conv3x3_winograd44s_lds_bytes(ctypes.byref(d)) == -1 and hip_lib.mr_conv3x3_winograd44s_f32(ctypes.byref(d), None) == -1


def test_host_geometry_shortcuts_are_bit_identical_to_the_reference_form():
    """model.host_geometry batches the F pose inversions into one ATen call and remembers the intrinsics inverse by content; both
    must reproduce the reference's one-call-per-matrix algebra (monorec_model.py:171,198,207) bit for bit on this host's CPU."""
    _check_host_geometry_against_the_reference_form()


@pytest.mark.gpu
def test_host_geometry_shortcuts_on_the_gpu_box_host(hip_lib):
    """The same statement on the GPU box's host CPU (another LAPACK / MKL code path than the build container's)."""
    _check_host_geometry_against_the_reference_form()


def test_output_relocation_table_of_a_plan(hip_lib):
    """Plan.rebind_outputs (forward(): outputs produced in caller-owned memory): every descriptor pointer into an output buffer
    moves with its buffer's new base, byte offsets kept (the mask encoder reads frame f of `sfcv` at f * B * D * H * W), closures
    follow through Plan.ref, and rebinding back restores every pointer.  The plan is built on the CPU: building launches nothing."""
    from monorec_amd import engine
    m = MonoRecModel(cv_depth_steps=8)
    sd = synth.seeded_state_dict(m.state_dict(), 0)
    plan = engine.Plan(sd, 2, 64, 96, 2, 8, (0.33, 0.0025), "cpu")
    assert plan.outputs_rebindable and set(plan.bound) == set(engine.OUTPUT_BUFFERS)

    def slots():
        out = []
        for obj, field, idx, name, off in plan._relocs:
            out.append(obj[idx] if field is None else (getattr(obj, field) if idx is None else getattr(obj, field)[idx]))
        return out
    names = [r[3] for r in plan._relocs]
    # writers and readers: conv1 writes feat0, the mask / depth decoders read it; the depth encoder reads the fused volume; the four
    # heads write the predictions; the cost-volume launch gets one pointer per frame of the single-frame volumes
    assert names.count("feat0") >= 3 and names.count("cost_volume") >= 1 and all(names.count(f"pred{i}") == 1 for i in range(4))
    assert names.count("sfcv") == 1 + 2
    resident = slots()
    bases = {n: 0x7000000000 + i * 0x10000000 for i, n in enumerate(plan.bound)}
    ref_cv, ref_kf = plan.ref(plan.buf["cost_volume"][1]), plan.ref(plan.buf["keyframe"])
    plan.rebind_outputs(bases)
    for (obj, field, idx, name, off), was, now in zip(plan._relocs, resident, slots()):
        assert was == plan._resident[name] + off and now == bases[name] + off
    assert ref_cv.ptr() == bases["cost_volume"] + 8 * 64 * 96 * 4 and ref_kf.ptr() == plan.buf["keyframe"].data_ptr()
    frame_offsets = sorted(off for _, field, _, name, off in plan._relocs if name == "sfcv" and field is None)
    assert frame_offsets == [0, 2 * 8 * 64 * 96 * 4]
    plan.rebind_outputs({"pred0": bases["pred0"]})            # the others fall back to the resident buffers
    assert plan.bound["pred0"] == bases["pred0"] and plan.bound["cost_volume"] == plan._resident["cost_volume"]
    plan.rebind_outputs(None)
    assert slots() == resident and plan.bound == plan._resident
    # option variants that keep constant content in an output buffer are not rebindable: forward() copies out of them instead
    m1 = MonoRecModel(cv_depth_steps=8, pretrain_mode=1)
    p1 = engine.Plan(synth.seeded_state_dict(m1.state_dict(), 0), 1, 64, 96, 2, 8, (0.33, 0.0025), "cpu", pretrain_mode=1)
    p2 = engine.Plan(sd, 1, 64, 96, 2, 8, (0.33, 0.0025), "cpu", no_cv=True)
    assert not p1.outputs_rebindable and not p2.outputs_rebindable


def test_host_wait_polls_for_a_bounded_time_then_sleeps():
    from monorec_amd import model as mm

    class Ev:
        def __init__(self, ready_after):
            self.n, self.ready_after, self.synced = 0, ready_after, False

        def query(self):
            self.n += 1
            return self.n > self.ready_after

        def synchronize(self):
            self.synced = True
    e = Ev(0)
    mm._host_wait(e)
    assert e.n == 1 and not e.synced                         # already done: one query
    e = Ev(50)
    mm._host_wait(e)
    assert e.n == 51 and not e.synced                        # short wait: polled
    old = mm.HOST_SPIN_SECONDS
    mm.HOST_SPIN_SECONDS = 0.0
    try:
        e = Ev(10 ** 9)
        mm._host_wait(e)
        assert e.synced and e.n <= 3                         # beyond the bound: handed to hipEventSynchronize
    finally:
        mm.HOST_SPIN_SECONDS = old


def test_forward_slot_avoids_uncollected_handles_of_copying_plans():
    """forward() keeps slot 0 unless the plan there copies out of its resident buffers AND a submit() handle of that slot is still
    uncollected."""
    import types
    import weakref
    from monorec_amd.model import _Pending
    m = MonoRecModel(cv_depth_steps=8, hip_in_flight=2)
    prep = types.SimpleNamespace(shape=(1, 64, 96, 2), device="cuda:0")
    assert m._forward_slot(prep) == 0                        # no plan yet
    key = lambda slot: (slot, 1, 64, 96, 2, 8, "cuda:0")
    h = _Pending({}, None, "cuda:0")
    m._plans[key(0)] = types.SimpleNamespace(outputs_rebindable=True, handles=[weakref.ref(h)])
    assert m._forward_slot(prep) == 0                        # outputs go to caller-owned memory: the handle's views are safe
    m._plans[key(0)].outputs_rebindable = False
    assert m._forward_slot(prep) == 1
    m._plans[key(1)] = types.SimpleNamespace(outputs_rebindable=False, handles=[weakref.ref(h)])
    with pytest.raises(RuntimeError, match="result has not been taken"):
        m._forward_slot(prep)
    h.collected = True
    assert m._forward_slot(prep) == 0


# ------------------------------------------------------------------------------------------ bf16 MFMA path with B8 activation storage
def test_b8_conv_desc_layout_matches_the_c_struct():
    fields = ", ".join(f'offsetof(mr_b8_conv_desc, {n})' for n, _ in _lib.B8ConvDesc._fields_)
    src = f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{ size_t o[] = {{{fields}}};' \
          'printf("%zu", sizeof(mr_b8_conv_desc)); for (unsigned i = 0; i < sizeof(o)/sizeof(o[0]); ++i) printf(" %zu", o[i]); return 0; }'
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(c, "w").write(src)
        subprocess.run(["gcc", c, "-o", exe], check=True)
        vals = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_lib.B8ConvDesc)
    assert vals[1:] == [getattr(_lib.B8ConvDesc, n).offset for n, _ in _lib.B8ConvDesc._fields_]


def test_b8_weight_packing(hip_lib):
    """mr_b8_pack_weights: the bf16 A-fragment stream of csrc/conv_b8.hip - [cout group of 16 mb][chunk of 32 channels, source-major][tap]
    [cout block][64 lanes][8]; lane l = (cout l & 15, channel block l >> 4 of the chunk), element e = channel 8 (l >> 4) + e of the chunk;
    chunks never straddle sources; padded channels / couts are zero; values rounded to bf16, nearest even."""
    g = torch.Generator().manual_seed(21)
    srcs_c, cout, kh, kw, mb = [5, 44], 40, 3, 2, 2
    cin, taps = sum(srcs_c), kh * kw
    w = torch.randn(cout, cin, kh, kw, generator=g)
    sc = (ctypes.c_int32 * len(srcs_c))(*srcs_c)
    nch = [((c + 7) // 8 + 3) // 4 for c in srcs_c]                  # 1 + 2 chunks
    groups = ((cout + 15) // 16 + mb - 1) // mb
    n = hip_lib.mr_b8_packed_weight_bytes(cout, sc, len(srcs_c), kh, kw, mb)
    assert n == groups * sum(nch) * taps * mb * 1024
    packed = torch.empty(n // 2, dtype=torch.bfloat16)
    _lib.check(hip_lib.mr_b8_pack_weights(w.data_ptr(), cout, sc, len(srcs_c), kh, kw, mb, packed.data_ptr()))
    st = packed.float().view(groups, sum(nch), taps, mb, 64, 8)
    wb = w.to(torch.bfloat16).float().view(cout, cin, taps)
    q0, cin_off = 0, 0
    for c, nq in zip(srcs_c, nch):
        for q in range(nq):
            for lane in range(64):
                for e in range(8):
                    cl = q * 32 + (lane >> 4) * 8 + e
                    for gi in range(groups):
                        for m in range(mb):
                            co = (gi * mb + m) * 16 + (lane & 15)
                            exp = wb[co, cin_off + cl] if (co < cout and cl < c) else torch.zeros(taps)
                            assert torch.equal(st[gi, q0 + q, :, m, lane, e], exp), (co, cl)
        q0, cin_off = q0 + nq, cin_off + c
    assert hip_lib.mr_b8_pack_weights(w.data_ptr(), cout, sc, len(srcs_c), kh, kw, 5, packed.data_ptr()) == -1


def test_bf16_mode_plan_stores_mask_and_depth_activations_in_b8(hip_lib, monkeypatch):
    """hip_bf16=True: the mask and depth nets run on mr_conv2d_b8 with B8 activations between their convolutions; dense fp32 stay the
    path's outputs and the maps the one-channel kernels read; MR_B8=0 and the option variants keep the fp32-storage path."""
    m = MonoRecModel(cv_depth_steps=32)
    sd = synth.seeded_state_dict(m.state_dict())
    plan = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", bf16=1)
    b8 = [c for c in plan.conv_log if c.get("b8")]
    assert plan.b8 and len(b8) == 53 and len(plan.conv_log) == 73 and all(c["name"].startswith("resnet.") for c in plan.conv_log if not c.get("b8"))
    assert abs(plan.conv_ref_macs() / 1e9 - (61.07 - 0.068)) < 0.01          # the reference's conv multiply-adds, as in the fp32 plan
    by = {c["name"]: c for c in b8}
    assert by["mask.enc0.0"]["spec"]["src_layouts"] == [1] and by["mask.enc0.0"]["spec"]["out_layout"] == 1       # the B8 copy the cost-volume fusion kernel writes
    assert by["mask.dec1.1"]["spec"]["src_layouts"] == [1, 1, 1]                                                   # B8 / B8 copy of the image features (round 6) / B8
    assert [n for n, _ in plan.stages["main"]][:4] == [f"feat{i}.to_b8" for i in range(4)] and tuple(plan.buf["feat0_b8"].shape) == (1, 8, 128, 256, 8)
    assert by["depth.dec3"]["spec"]["src_layouts"] == [1, 1, 0]                                                    # ... the fp32 map the depth heads read stays fp32
    assert by["depth.enc0.0.conv_y"]["spec"]["src_layouts"] == [1, 0]                                              # B8 copy of the masked volume (classifier kernel) + keyframe
    assert plan._sfcv_b8_ptrs is not None and tuple(plan.buf["sfcv_b8"].shape) == (2, 4, 256, 512, 8) and tuple(plan.buf["cost_volume_b8"].shape) == (1, 4, 256, 512, 8)
    fp32_out = {c["name"] for c in b8 if c["spec"]["out_layout"] == 0}
    assert fp32_out == {"mask.dec3.2", "depth.dec0", "depth.dec1.1.conv_x", "depth.dec2.1.conv_x", "depth.dec4.2"}     # classifier / head inputs
    assert all(0 < c["lds"] <= 160 * 1024 for c in b8)
    assert plan.buf["mask.enc0.x"].dtype == torch.bfloat16 and tuple(plan.buf["mask.enc0.x"].shape) == (2, 4, 256, 512, 8)
    assert plan.buf["cost_volume"].dtype == torch.float32 and plan.buf["feat2"].dtype == torch.float32 and plan.buf["pred0"].dtype == torch.float32
    assert plan.outputs_rebindable and {r[3] for r in plan._relocs} >= {"sfcv", "feat0", "feat1", "feat2", "pred0", "pred3"}    # (the fused volume: closures only)
    monkeypatch.setenv("MR_B8_FEATS", "0")                                                                         # A/B aid: the decoders stage the fp32 features (rounds 4-5)
    nofc = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", bf16=1)
    assert {c["name"]: c for c in nofc.conv_log}["mask.dec1.1"]["spec"]["src_layouts"] == [1, 0, 1] and "feat0_b8" not in nofc.buf
    monkeypatch.delenv("MR_B8_FEATS")
    monkeypatch.setenv("MR_B8", "0")
    old = engine.Plan(sd, 1, 256, 512, 2, 32, (0.33, 0.0025), "cpu", bf16=1)
    assert not old.b8 and not any(c.get("b8") for c in old.conv_log) and all(c["bf16"] == 1 for c in old.conv_log)
    monkeypatch.delenv("MR_B8")
    var = engine.Plan(sd, 1, 64, 96, 2, 32, (0.33, 0.0025), "cpu", bf16=1, mask_use_feats=False)
    assert not var.b8 and not any(c.get("b8") for c in var.conv_log)


def test_lds_plane_pitches_are_conflict_free_for_the_width_of_their_reads():
    """tools/lds_banks.py restates the lane groups / bank moduli of MI355X_MICROARCH.md (LDS): the plane pitches compiled into the kernels
    (read back from the sources) serve their patch reads in the minimum number of LDS cycles; the pitches of rounds 3-4 took twice as many."""
    import sys
    sys.path.insert(0, ROOT)
    from tools import lds_banks
    src = lambda f: open(os.path.join(ROOT, "monorec_amd", "csrc", f)).read()
    w44 = src("conv_wino44.hip")
    rows, pitch = int(re.search(r"RH = (\d+)", w44).group(1)) + 2, int(re.search(r"RW = (\d+)", w44).group(1)) + 8
    plane44 = (rows * pitch + 63) // 64 * 64
    assert "constexpr int PLANE = (ROWS * PITCH + 63) / 64 * 64;" in w44 and plane44 == 1344
    b128 = lambda plane: lds_banks.cycles(lambda l: (l >> 4) * plane + 4 * (l & 15), 4)[0]
    assert b128(plane44) == 4 and b128(1296) == 8 and b128(576) == 4 and b128(592) == 8 and b128(768) == 4
    wino = src("conv_wino.hip")
    assert "constexpr int RAW_PLANE = 416;" in wino and "RAW_PLANE_T = 736;" in wino and "constexpr int RAW_PLANE_X = 416;" in src("conv1d_wino.hip")
    b64 = lambda plane: lds_banks.cycles(lambda l: (l >> 4) * plane + 2 * (l & 15) + 2, 2)[0]
    b32 = lambda plane: lds_banks.cycles(lambda l: (l >> 4) * plane + 2 * (l & 15) + 3, 1)[0]
    assert b64(416) == 2 and b64(736) == 2 and b64(400) == 4 and b32(400) == 4 and b32(416) == 4      # lane stride 2: dwords collide on any even pitch
    assert lds_banks.cycles(lambda l: (l >> 4) * 400 + (l & 15), 1)[0] == 2                              # dword reads at lane stride 1: 16 mod 32
    assert lds_banks.cycles(lambda l: 6 * l, 2)[0] == 2                                                  # conv_wino44 A operands: lane pitch 24 bytes


def test_b8_schedule_table_overrides_the_rule_per_launch_signature(monkeypatch):
    """engine.B8_SCHEDULES (tuned_b8.json, tools/tune_b8.py): a measured (MB, NB, waves) per mr_conv2d_b8 launch signature goes ahead of the rule
    Plan.b8_schedule; launches without an entry keep the rule's choice; the plan stamp follows the schedules."""
    m = MonoRecModel(cv_depth_steps=8)
    sd = synth.seeded_state_dict(m.state_dict(), seed=0)
    monkeypatch.setattr(engine, "B8_SCHEDULES", {})
    base = engine.Plan(sd, 1, 64, 128, 2, 8, (0.33, 0.0025), "cpu", bf16=1)
    b8 = [c for c in base.conv_log if c.get("b8")]
    assert b8 and all("f32_source" in c for c in b8)
    tgt = next(c for c in b8 if c["phases"] == 1 and (c["mb"], c["nb"], c["waves"]) != (1, 1, 4))
    key = tgt["sig"] + f"_f{int(tgt['f32_source'])}"
    monkeypatch.setattr(engine, "B8_SCHEDULES", {key: (1, 1, 4)})
    plan = engine.Plan(sd, 1, 64, 128, 2, 8, (0.33, 0.0025), "cpu", bf16=1)
    for c0, c1 in zip(b8, [c for c in plan.conv_log if c.get("b8")]):
        same_key = c0["sig"] + f"_f{int(c0['f32_source'])}" == key
        assert (c1["mb"], c1["nb"], c1["waves"]) == ((1, 1, 4) if same_key else (c0["mb"], c0["nb"], c0["waves"])), c0["name"]
    assert plan.launch_stamp() != base.launch_stamp()
