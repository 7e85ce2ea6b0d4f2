"""CPU emulation of the MR_COMPUTE_BF16 / MR_COMPUTE_BF16X3 convolution arithmetic over the whole network (diagnostic, not a test):

    python tests/diagnostics/emulate_bf16x3.py [H W D]        # default 256 512 32 = BASELINE configs[1]

Every conv / transposed conv of the oracle is evaluated on bf16-rounded operands (bf16), on the three-term split
x_hi*w_hi + x_hi*w_lo + x_lo*w_hi with hi = bf16(v), lo = bf16(v - hi) (bf16x3 - what conv_mfma_kernel<..., 2> computes), or with
the fourth term too (bf16x4); products of bf16 values are exact in fp32 and the accumulation is fp32, like the MFMA.  Measured on
the build container: c2 depth error bf16 2.2e-3, bf16x3 3.9e-6, bf16x4 2.8e-6 (fp32 HIP path vs CPU: 1.3e-6; parity bar 1e-4)."""
import torch, torch.nn.functional as F, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from monorec_amd import synth, MonoRecModel
from oracle import monorec_oracle as orc

def split(t):
    hi = t.to(torch.bfloat16).float()
    lo = (t - hi).to(torch.bfloat16).float()
    return hi, lo

orig_conv2d, orig_convT = F.conv2d, F.conv_transpose2d
MODE = {"m": "fp32"}
def conv2d(x, w, b=None, *a, **k):
    if MODE["m"] == "fp32" or x.dtype != torch.float32:
        return orig_conv2d(x, w, b, *a, **k)
    xh, xl = split(x); wh, wl = split(w)
    if MODE["m"] == "bf16":
        return orig_conv2d(xh, wh, b, *a, **k)
    y = orig_conv2d(xh, wh, b, *a, **k) + orig_conv2d(xh, wl, None, *a, **k) + orig_conv2d(xl, wh, None, *a, **k)
    if MODE["m"] == "bf16x4":
        y = y + orig_conv2d(xl, wl, None, *a, **k)
    return y
def convT(x, w, b=None, *a, **k):
    if MODE["m"] == "fp32":
        return orig_convT(x, w, b, *a, **k)
    xh, xl = split(x); wh, wl = split(w)
    if MODE["m"] == "bf16":
        return orig_convT(xh, wh, b, *a, **k)
    y = orig_convT(xh, wh, b, *a, **k) + orig_convT(xh, wl, None, *a, **k) + orig_convT(xl, wh, None, *a, **k)
    if MODE["m"] == "bf16x4":
        y = y + orig_convT(xl, wl, None, *a, **k)
    return y

h, w, d = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 512, 32)
m = MonoRecModel(cv_depth_steps=d)
sd = synth.seeded_state_dict(m.state_dict(), seed=0)
batch = synth.make_batch(1, h, w, 2, seed=1)
# cost volume once (not a conv), network under the different conv arithmetic
ref = orc.forward(sd, batch, cv_depth_steps=d)
F.conv2d, F.conv_transpose2d = conv2d, convT
try:
    for mode in ("bf16", "bf16x3", "bf16x4"):
        MODE["m"] = mode
        out = orc.forward(sd, batch, cv_depth_steps=d)
        e = (out["result"] - ref["result"]).abs()
        em = (out["cv_mask"] - ref["cv_mask"]).abs()
        print(f"{mode:7s} result max|err| {e.max().item():.3e}  mean {e.mean().item():.3e}   cv_mask max {em.max().item():.3e}   (result range {ref['result'].min().item():.4f}..{ref['result'].max().item():.4f})")
finally:
    F.conv2d, F.conv_transpose2d = orig_conv2d, orig_convT
