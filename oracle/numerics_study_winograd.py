"""TEST INFRASTRUCTURE (CPU only, never imported by the product): what larger Winograd / Cook-Toom tiles would do to the parity bar.

The product first ran the 3x3 stride-1 layers on F(2x2,3x3) and the 3x1 / 1x3 layers on F(2,3): transform
coefficients 0, +-1, +-1/2, depth within 1.3e-6 of the CPU output.  The next step in multiplies is F(4x4,3x3) (36 instead of 64
multiplies per 4x4 outputs), F(4,3) along one axis (6 instead of 8 per 4 outputs) and F(2,7) / F(4,7) for the 7-tap layers of
DepthModule.enc.0.0 (model/monorec/monorec_model.py:487-500, model/layers.py:289-314) - with interpolation points +-2, +-1/2, +-4 and
constants up to 89 and 1/2835.  BEFORE any of those kernels was written this script answered, in emulated fp32 on the oracle's own
activations: how far does `result` move when those layers are evaluated that way?  (bar: 1e-4 on the depth; answer: 2-4e-7,
profiles/r03_numerics_study_winograd.json; kept alive by tests/test_numerics_gate.py)

    python -m oracle.numerics_study_winograd [--height 256 --width 512] [--json out.json]

The emulation follows the kernels' structure: transformed weights formed in double and rounded once, input transform in fp32,
channel sum in fp32 (torch matmul; the MFMA's exact k-ordered chain differs from it by summation order only), output transform in fp32.
"""
import argparse
import json
import sys

import numpy as np
import torch
import torch.nn.functional as F

from monorec_amd import cooktoom, synth
from oracle import monorec_oracle as oracle


# ---------------------------------------------------------------------------------------------------------------- Cook-Toom matrices
# (ONE derivation for the study, the generated kernel header and the tests: monorec_amd/cooktoom.py, exact rational arithmetic)
_CACHE = {}


def matrices(m, r):
    """(A^T, G, B^T) of F(m, r) as float64 arrays."""
    if (m, r) not in _CACHE:
        _CACHE[(m, r)] = tuple(np.array([[float(v) for v in row] for row in mat], dtype=np.float64) for mat in cooktoom.cook_toom(m, r))
    return _CACHE[(m, r)]


# ---------------------------------------------------------------------------------------------------------------- emulated convolutions
def winograd_1d(x, w, bias, axis, m):
    """stride-1 'same' correlation with a (r x 1) [axis 2] or (1 x r) [axis 3] filter as F(m, r) in emulated fp32."""
    r = w.shape[axis]
    at, g, bt = matrices(m, r)
    n = m + r - 1
    if axis == 2:
        return winograd_1d(x.transpose(2, 3), w.transpose(2, 3), bias, 3, m).transpose(2, 3)
    pl, pr = oracle.same_pad(x.shape[3], r, 1)
    width = x.shape[3]
    tiles = -(-width // m)
    xp = F.pad(x, [pl, tiles * m + r - 1 - width - pl, 0, 0])
    d = xp.unfold(3, n, m)                                                    # N C H T n
    u = torch.einsum("ij,ocj->oci", torch.from_numpy(g), w[:, :, 0, :].double()).float()          # O C n (rounded once)
    v = torch.einsum("ij,nchtj->nchti", torch.from_numpy(bt).float(), d)     # fp32 input transform
    mm = torch.einsum("oci,nchti->nohti", u, v)                                 # fp32 channel sum per position
    y = torch.einsum("ki,nohti->nohtk", torch.from_numpy(at).float(), mm)     # fp32 output transform
    y = y.reshape(y.shape[0], y.shape[1], y.shape[2], tiles * m)[..., :width]
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def _cooktoom_valid(xp, w3, m):
    """'valid' stride-1 correlation along the LAST axis as F(m, r) in emulated fp32: xp (N, C, H, L), w3 (O, C, r) -> (N, O, H, L - r + 1)."""
    r = w3.shape[2]
    out_len = xp.shape[3] - r + 1
    if r == 1:                                                   # one tap: a plain product, no transform
        return torch.einsum("oc,nchl->nohl", w3[:, :, 0], xp[..., :out_len])
    at, g, bt = matrices(m, r)
    n = m + r - 1
    tiles = -(-out_len // m)
    xq = F.pad(xp, [0, tiles * m + r - 1 - xp.shape[3], 0, 0])
    d = xq.unfold(3, n, m)                                                     # N C H T n
    u = torch.einsum("ij,ocj->oci", torch.from_numpy(g), w3.double()).float()  # rounded once
    v = torch.einsum("ij,nchtj->nchti", torch.from_numpy(bt).float(), d)
    mm = torch.einsum("oci,nchti->nohti", u, v)
    y = torch.einsum("ki,nohti->nohtk", torch.from_numpy(at).float(), mm)
    return y.reshape(y.shape[0], y.shape[1], y.shape[2], tiles * m)[..., :out_len]


def cooktoom_1d_stride2(x, w, bias, axis, m):
    """A (r x 1) [axis 2] / (1 x r) [axis 3] correlation with stride 2 along the filter axis and TF-'same' padding (layers.ConvReLU2 with stride 2,
    reference model/layers.py:289-314) as the POLYPHASE pair of monorec_amd.cooktoom.polyphase_stride2: even taps on the even samples + odd taps on
    the odd samples of the padded input, each a stride-1 F(m, .) in emulated fp32, the two outputs added in fp32."""
    if axis == 2:
        return cooktoom_1d_stride2(x.transpose(2, 3), w.transpose(2, 3), bias, 3, m).transpose(2, 3)
    r = w.shape[3]
    pl, pr = oracle.same_pad(x.shape[3], r, 2)
    n_out = -(-x.shape[3] // 2)
    xp = F.pad(x, [pl, pr + 2 * (m + r), 0, 0])                                   # generous zero tail: the tiles run past the last output
    ev, od = cooktoom.polyphase_stride2(r, pl)
    y = None
    for samples, taps in ((xp[..., 0::2], ev), (xp[..., 1::2], od)):
        if not taps:
            continue
        part = _cooktoom_valid(samples.contiguous(), w[:, :, 0, taps], m)[..., :n_out]
        y = part if y is None else y + part
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def cooktoom_1d_stride2_unified(x, w, bias, axis, m=4):
    """What the product's stride-2 kernels compute (round 5, csrc/conv1d_wino.hip + engine.Plan._conv_relu2_stride2): the r-tap stride-2
    correlation as ONE ceil(r/2)-tap stride-1 Cook-Toom form over the channel concatenation [even samples | odd samples] of the input
    (monorec_amd.cooktoom.stride2_as_stride1: 7 taps -> F(m,4), 5 taps -> F(m,3); the missing tap of the shorter phase is a zero weight), in
    emulated fp32.  Needs an even input length along the filter axis (every stride-2 layer of the DepthModule has one)."""
    if axis == 2:
        return cooktoom_1d_stride2_unified(x.transpose(2, 3), w.transpose(2, 3), bias, 3, m).transpose(2, 3)
    r, n, c = w.shape[3], x.shape[3], x.shape[1]
    r2, pad, ev, od = cooktoom.stride2_as_stride1(r, n)
    xe = torch.cat([x[..., 0::2], x[..., 1::2]], 1)
    w2 = torch.zeros(w.shape[0], 2 * c, r2, dtype=w.dtype)
    for t in range(r2):
        if ev[t] is not None:
            w2[:, :c, t] = w[:, :, 0, ev[t]]
        if od[t] is not None:
            w2[:, c:, t] = w[:, :, 0, od[t]]
    n_out = n // 2
    xp = F.pad(xe, [pad, r2 - 1 - pad + m + r2, 0, 0])
    y = _cooktoom_valid(xp.contiguous(), w2, m)[..., :n_out]
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def winograd_2d(x, w, bias, m):
    """3x3 stride-1 'same' convolution as F(m x m, 3 x 3) in emulated fp32."""
    r = 3
    at, g, bt = matrices(m, r)
    n = m + r - 1
    h, wd = x.shape[2], x.shape[3]
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x, [1, tw * m + 2 - wd - 1, 1, th * m + 2 - h - 1])
    d = xp.unfold(2, n, m).unfold(3, n, m)                                      # N C TH TW n n
    g64 = torch.from_numpy(g)
    u = torch.einsum("ij,ocjk,lk->ocil", g64, w.double(), g64).float()          # O C n n
    btf, atf = torch.from_numpy(bt).float(), torch.from_numpy(at).float()
    v = torch.einsum("ij,ncabjk->ncabik", btf, d)
    v = torch.einsum("ncabik,lk->ncabil", v, btf)
    out = x.new_empty(x.shape[0], w.shape[0], th, tw, m, m)
    for a0 in range(0, th, 8):                                                   # bounded temporaries
        mm = torch.einsum("ocil,ncabil->noabil", u, v[:, :, a0:a0 + 8])
        y = torch.einsum("pi,noabil->noabpl", atf, mm)
        out[:, :, a0:a0 + 8] = torch.einsum("noabpl,ql->noabpq", y, atf)
    y = out.permute(0, 1, 2, 4, 3, 5).reshape(x.shape[0], w.shape[0], th * m, tw * m)[:, :, :h, :wd]
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


# ---------------------------------------------------------------------------------------------------------------- the experiment
class Patched:
    """oracle.conv_same replaced for the layers `rule(weight_shape, stride, input_shape)` selects (returns None or m)."""

    def __init__(self, rule):
        self.rule, self.hits, self.layer_err = rule, 0, []

    def __enter__(self):
        self.orig = oracle.conv_same
        outer = self

        def conv_same(x, w, b, stride=(1, 1)):
            m = outer.rule(tuple(w.shape), tuple(stride), tuple(x.shape))
            if m is None:
                return outer.orig(x, w, b, stride)
            kh, kw = w.shape[2], w.shape[3]
            if tuple(stride) != (1, 1) and m < 0:                # the unified [even | odd] form the product runs (F(-m, ceil(taps / 2)))
                y = cooktoom_1d_stride2_unified(x, w, b, 2 if kw == 1 else 3, -m)
            elif tuple(stride) != (1, 1):                        # (2, 1) with a k x 1 filter or (1, 2) with a 1 x k filter
                y = cooktoom_1d_stride2(x, w, b, 2 if kw == 1 else 3, m)
            else:
                y = winograd_2d(x, w, b, m) if (kh, kw) == (3, 3) else winograd_1d(x, w, b, 2 if kw == 1 else 3, m)
            ref = outer.orig(x.double(), w.double(), None if b is None else b.double(), stride)
            direct = outer.orig(x, w, b, stride)
            outer.hits += 1
            outer.layer_err.append(dict(shape=list(w.shape), m=m, scale=float(ref.abs().max()),
                                        winograd_vs_fp64=float((y.double() - ref).abs().max()),
                                        direct_vs_fp64=float((direct.double() - ref).abs().max())))
            return y
        oracle.conv_same = conv_same
        return self

    def __exit__(self, *exc):
        oracle.conv_same = self.orig


def rules():
    def only(kh, kw, m, min_pixels=0):
        def rule(ws, stride, xs):
            if (ws[2], ws[3]) == (kh, kw) and stride == (1, 1) and ws[0] > 1 and xs[2] * xs[3] >= min_pixels:   # (one-channel heads: csrc/heads.hip)
                return m
            return None
        return rule

    def stride2(m):                                             # the stride-2 halves of ConvReLU2: 7 / 5 / 3 taps along the strided axis
        def rule(ws, stride, xs):
            if (stride == (2, 1) and ws[3] == 1 and ws[2] in (3, 5, 7)) or (stride == (1, 2) and ws[2] == 1 and ws[3] in (3, 5, 7)):
                return m
            return None
        return rule

    def stride2_unified(m):                                     # what the product runs: the 7- and 5-tap stride-2 layers only (3 taps: no gain)
        def rule(ws, stride, xs):
            if (stride == (2, 1) and ws[3] == 1 and ws[2] in (5, 7) and xs[2] % 2 == 0) or (stride == (1, 2) and ws[2] == 1 and ws[3] in (5, 7) and xs[3] % 2 == 0):
                return -m
            return None
        return rule

    def both(*rs):
        def rule(ws, stride, xs):
            for r_ in rs:
                m = r_(ws, stride, xs)
                if m is not None:
                    return m
            return None
        return rule
    return {
        "F(2x2,3x3) [what the product runs]": only(3, 3, 2),
        "F(4x4,3x3) all 3x3 stride-1 layers": only(3, 3, 4),
        "F(4x4,3x3) layers of >= 64x128 pixels only": only(3, 3, 4, 64 * 128),
        "F(2,3) 3x1 + 1x3 [what the product runs]": both(only(3, 1, 2), only(1, 3, 2)),
        "F(4,3) 3x1 + 1x3": both(only(3, 1, 4), only(1, 3, 4)),
        "F(2,7) 7x1 + 1x7": both(only(7, 1, 2), only(1, 7, 2)),
        "F(4,7) 7x1 + 1x7": both(only(7, 1, 4), only(1, 7, 4)),
        "F(4x4,3x3) + F(4,3) + F(2,7) together": both(only(3, 3, 4), only(3, 1, 4), only(1, 3, 4), only(7, 1, 2), only(1, 7, 2)),
        "polyphase F(2,.) stride-2 k x 1 / 1 x k": stride2(2),
        "polyphase F(4,.) stride-2 k x 1 / 1 x k": stride2(4),
        "unified polyphase F(4,4) / F(4,3) stride-2 7- and 5-tap layers [what the product runs]": stride2_unified(4),
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--depths", type=int, default=32)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="substring of the experiment name")
    ap.add_argument("--family", default="he", choices=synth.WEIGHT_FAMILIES, help="weight family of synth.seeded_state_dict")
    args = ap.parse_args(argv)
    torch.manual_seed(0)
    batch = synth.make_batch(1, args.height, args.width, 2, seed=1)
    from monorec_amd.model import MonoRecModel   # state-dict layout only (CPU construction, no launch)
    sd = synth.seeded_state_dict(MonoRecModel(cv_depth_steps=args.depths).state_dict(), 0, args.family)
    base = oracle.forward(sd, synth.clone_batch(batch), cv_depth_steps=args.depths)
    report = {"shape": [args.height, args.width, args.depths], "bar": 1e-4, "family": args.family, "experiments": {}}
    for name, rule in rules().items():
        if args.only and args.only not in name:
            continue
        with Patched(rule) as p:
            out = oracle.forward(sd, synth.clone_batch(batch), cv_depth_steps=args.depths)
        err = float((out["result"] - base["result"]).abs().max())
        mask_err = float((out["cv_mask"] - base["cv_mask"]).abs().max())
        worst = max(p.layer_err, key=lambda e: e["winograd_vs_fp64"] / max(e["scale"], 1e-30)) if p.layer_err else None
        report["experiments"][name] = dict(layers=p.hits, result_max_abs_diff=err, cv_mask_max_abs_diff=mask_err, worst_layer=worst,
                                           median_layer_err=float(np.median([e["winograd_vs_fp64"] for e in p.layer_err])) if p.layer_err else None,
                                           median_direct_err=float(np.median([e["direct_vs_fp64"] for e in p.layer_err])) if p.layer_err else None)
        print(f"{name:48s} layers {p.hits:2d}  result {err:.2e}  cv_mask {mask_err:.2e}  "
              f"layer err (median) {report['experiments'][name]['median_layer_err']:.2e} vs direct {report['experiments'][name]['median_direct_err']:.2e}",
              flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)
    return report


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
