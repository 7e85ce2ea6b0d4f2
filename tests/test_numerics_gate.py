"""CPU: the numerics gate of the reduced-multiply kernels.  The larger Cook-Toom / Winograd forms (F(4,3), F(2,7), F(4,7), F(4x4,3x3):
csrc/conv1d_wino.hip, csrc/conv_wino44.hip) have transform coefficients up to 89 and thirds / 2835ths in G; before any of those kernels was
written, oracle/numerics_study_winograd.py measured on the oracle's own forward what evaluating the affected layers that way does to `result`
(emulated fp32, transformed weights rounded once from double).  This test keeps that measurement alive at a small shape: every form the
product's table may select must leave the depth far inside the 1e-4 parity bar."""
import pytest

from oracle import numerics_study_winograd as study


def test_every_form_the_table_may_select_keeps_the_depth_far_inside_the_parity_bar():
    report = study.main(["--height", "64", "--width", "128", "--depths", "32"])
    exps = report["experiments"]
    assert len(exps) == 11 and report["bar"] == 1e-4          # 8 forms the product carries + the two polyphase stride-2 candidates of round 4 + the unified
                                                               # [even | odd] form the round-5 stride-2 kernels run
    for name, e in exps.items():
        assert e["layers"] > 0, name
        assert e["result_max_abs_diff"] <= 2e-6, (name, e["result_max_abs_diff"])          # measured 1.5e-7 .. 2.8e-7: 1/50 of this bound, 1/350 of the bar
        assert e["cv_mask_max_abs_diff"] <= 1e-5, (name, e["cv_mask_max_abs_diff"])
    # per layer the larger forms do round more than the direct sum - the reason the gate exists
    assert exps["F(4,7) 7x1 + 1x7"]["median_layer_err"] > 5 * exps["F(4,7) 7x1 + 1x7"]["median_direct_err"]
    assert exps["F(4x4,3x3) all 3x3 stride-1 layers"]["median_layer_err"] > 2 * exps["F(4x4,3x3) all 3x3 stride-1 layers"]["median_direct_err"]


def test_cook_toom_emulation_equals_the_direct_convolution_up_to_rounding():
    import math
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 13, 22, generator=g)
    for m, r, tol in ((2, 3, 1e-5), (4, 3, 2e-5), (2, 7, 3e-5), (4, 7, 3e-4)):
        for axis in (2, 3):
            kk = (r, 1) if axis == 2 else (1, r)
            w = torch.randn(5, 9, *kk, generator=g) / math.sqrt(9.0 * r)
            b = torch.randn(5, generator=g)
            ref = F.conv2d(x, w, b, padding=(kk[0] // 2, kk[1] // 2))
            assert float((study.winograd_1d(x, w, b, axis, m) - ref).abs().max()) <= tol, (m, r, axis)
    w = torch.randn(5, 9, 3, 3, generator=g) / 9.0
    for m, tol in ((2, 1e-5), (4, 5e-5)):
        assert float((study.winograd_2d(x, w, None, m) - F.conv2d(x, w, None, padding=1)).abs().max()) <= tol, m


def test_polyphase_stride2_emulation_equals_the_strided_convolution_up_to_rounding():
    """oracle/numerics_study_winograd.cooktoom_1d_stride2 - the stride-2 halves of ConvReLU2 (7 / 5 / 3 taps, TF-'same' padding, reference
    model/layers.py:241-252,289-314) as even taps on even samples + odd taps on odd samples, each a stride-1 Cook-Toom form - against the oracle's
    own strided convolution, odd and even input lengths."""
    import math
    import torch
    from oracle import monorec_oracle as oracle
    g = torch.Generator().manual_seed(6)
    for h, wd in ((14, 22), (13, 21)):
        x = torch.randn(2, 7, h, wd, generator=g)
        for r in (3, 5, 7):
            for axis in (2, 3):
                kk, stride = ((r, 1), (2, 1)) if axis == 2 else ((1, r), (1, 2))
                w = torch.randn(6, 7, *kk, generator=g) / math.sqrt(7.0 * r)
                b = torch.randn(6, generator=g)
                ref = oracle.conv_same(x, w, b, stride)
                for m in (2, 4):
                    got = study.cooktoom_1d_stride2(x, w, b, axis, m)
                    assert got.shape == ref.shape and float((got - ref).abs().max()) <= 2e-5, (h, wd, r, axis, m, float((got - ref).abs().max()))


def test_unified_polyphase_emulation_equals_the_strided_convolution_up_to_rounding():
    """oracle/numerics_study_winograd.cooktoom_1d_stride2_unified - what the round-5 stride-2 kernels compute: the 7- / 5-tap stride-2 halves of
    ConvReLU2 (reference model/layers.py:241-252,289-314; monorec_model.py:489-501) as ONE 4- / 3-tap stride-1 Cook-Toom form F(4,4) / F(4,3) over
    the channel concatenation [even samples | odd samples] (monorec_amd.cooktoom.stride2_as_stride1) - against the oracle's strided convolution."""
    import math
    import torch
    from monorec_amd import cooktoom
    from oracle import monorec_oracle as oracle
    assert cooktoom.stride2_as_stride1(7, 256) == (4, 1, [0, 2, 4, 6], [1, 3, 5, None])
    assert cooktoom.stride2_as_stride1(5, 64) == (3, 1, [None, 1, 3], [0, 2, 4])
    g = torch.Generator().manual_seed(6)
    for h, wd in ((14, 22), (16, 24)):
        x = torch.randn(2, 7, h, wd, generator=g)
        for r in (5, 7):
            for axis in (2, 3):
                kk, stride = ((r, 1), (2, 1)) if axis == 2 else ((1, r), (1, 2))
                w = torch.randn(6, 7, *kk, generator=g) / math.sqrt(7.0 * r)
                b = torch.randn(6, generator=g)
                ref = oracle.conv_same(x, w, b, stride)
                got = study.cooktoom_1d_stride2_unified(x, w, b, axis, 4)
                assert got.shape == ref.shape and float((got - ref).abs().max()) <= 2e-5, (h, wd, r, axis, float((got - ref).abs().max()))
