"""Diagnostic (GPU box): the real KITTI sample with the fixture's own matrices - where, if anywhere, does the all-depth validity of
the HIP single-frame volumes differ from the reference's (tests/golden/kitti_example_169.npz: geom.valid_bits)?  For every flipped
pixel the oracle is evaluated on this host with the same matrices to show how marginal the border-mask sample is."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import Golden                                        # noqa: E402
from monorec_amd import synth                                         # noqa: E402
from monorec_amd.model import MonoRecModel                            # noqa: E402
from oracle import monorec_oracle as orc                              # noqa: E402

g = Golden("kitti_example_169")
batch = g.make_inputs()
kinv, proj = torch.from_numpy(g.z["geom.kinv"]), torch.from_numpy(g.z["geom.proj"])
m = MonoRecModel(cv_depth_steps=32, hip_in_flight=1)
sd = synth.seeded_state_dict(m.state_dict(), seed=0)
m.load_state_dict(sd)
m = m.to("cuda:0").eval()
m._geometry_override = (kinv, proj)
with torch.no_grad():
    out = m(synth.clone_batch(batch, "cuda:0"))
torch.cuda.synchronize()
sf = [s.cpu() for s in out["single_frame_cvs"]]
F_, H, W = len(sf), g.h, g.w
vb = np.unpackbits(g.z["geom.valid_bits"])[: F_ * H * W].reshape(F_, H, W).astype(bool)
hip_valid = np.stack([~(s[0] == 0).all(0).numpy() for s in sf])
flips = np.argwhere(hip_valid != vb)
print("validity flips HIP vs reference fixture:", len(flips), "of", vb.size)
# the oracle on THIS host with the fixture's matrices (projection_matrix / inverse patched)
orig_pm, orig_inv = orc.projection_matrix, torch.inverse
calls = {"f": 0}


def pm(src_intr, src_pose, kf_pose):
    f = calls["f"] % F_
    calls["f"] += 1
    return proj[0, f].view(1, 3, 4)


def inv(x):
    if x.shape == (4, 4) and torch.equal(x, batch["keyframe_intrinsics"][0]):
        k = torch.eye(4)
        k[:3, :3] = kinv[0].view(3, 3)
        return k
    return orig_inv(x)


orc.projection_matrix, torch.inverse = pm, inv
st = {}
ocv, osf = orc.cost_volume(batch, steps=32, stages=st)
orc.projection_matrix, torch.inverse = orig_pm, orig_inv
o_valid = torch.stack(st["valid"])[0].squeeze(1).numpy().astype(bool)
print("oracle(this host, fixture matrices) vs fixture flips:", int((o_valid != vb).sum()), " HIP vs oracle(this host):", int((hip_valid != o_valid).sum()))
for f in range(F_):
    d = (sf[f] - osf[f]).abs()
    print(f"frame {f}: sfcv HIP vs oracle(this host, fixture matrices): max {d.max():.3e}, frac>1e-5 {(d > 1e-5).float().mean():.2e}")
grid = st["grid"][0]                                                    # (F,D,H,W,2)
for f, y, x in flips[:20]:
    gx, gy = grid[f, :, y, x, 0], grid[f, :, y, x, 1]
    sx, sy = ((gx + 1) * W - 1) / 2, ((gy + 1) * H - 1) / 2
    print(f"flip f={f} y={y} x={x}: fixture valid={vb[f, y, x]} hip={hip_valid[f, y, x]} oracle_here={o_valid[f, y, x]}; "
          f"sample x range [{sx.min():.4f}, {sx.max():.4f}] y range [{sy.min():.4f}, {sy.max():.4f}]")
r = (out["result"].cpu() - torch.from_numpy(g.z["result.full"])).abs()
print("result vs fixture: max %.3e, frac>1e-4 %.4f" % (r.max(), (r > 1e-4).float().mean()))
cvm = (out["cost_volume"].cpu() - ocv * (1 - out["cv_mask"].cpu())).abs()
print("masked cost volume HIP vs oracle(this host, fixture matrices): max %.3e frac>1e-4 %.2e" % (cvm.max(), (cvm > 1e-4).float().mean()))

# ---- round 2, second pass: where does `result` leave the fixture although validity and the single-frame volumes agree? ----------
print("--- vs fixture, stage by stage (strided samples unless .full)")
for name, t in [("cost_volume", out["cost_volume"]), ("cv_mask", out["cv_mask"])] + [(f"feat{i}", out["image_features"][i]) for i in range(5)] + \
        [(f"pred{i}", out["predicted_inverse_depths"][i]) for i in range(4)]:
    if (name + ".samples") not in g.z.files:
        continue
    flat = t.detach().cpu().float().reshape(-1)
    got = flat[::int(g.z[name + ".stride"])].numpy()
    want = g.z[name + ".samples"]
    e = np.abs(got - want)
    print(f"{name:12s} max {e.max():.3e}  frac>1e-5 {np.mean(e > 1e-5):.2e}  frac>1e-4 {np.mean(e > 1e-4):.2e}  n {e.size}")
rd = (out["result"].cpu() - torch.from_numpy(g.z["result.full"])).abs()[0, 0]
ys, xs = np.unravel_index(np.argsort(rd.numpy().reshape(-1))[-5:], rd.shape)
print("largest result deviations at (y, x):", list(zip(ys.tolist(), xs.tolist())), [float(rd[y, x]) for y, x in zip(ys, xs)])
# the fusion formula of the reference (monorec_model.py:257-269) applied on the CPU to the sads recovered from the HIP volumes
hip_cv = out["cost_volume"].cpu() / (1 - out["cv_mask"].cpu()).clamp_min(1e-12)
valid_t = torch.from_numpy(hip_valid).unsqueeze(1).float()                      # (F,1,H,W)
sad_hip = torch.stack([(1 - s[0]) / 2 for s in sf])                              # (F,D,H,W); meaningful where valid
e_ = torch.exp(-10 * torch.pow(sad_hip - sad_hip.min(1, keepdim=True)[0], 2))
w_ = (1 - 1 / 31 * (e_.sum(1, keepdim=True) - 1)) * valid_t
c_ = (sad_hip * w_).sum(0)
S_ = w_.sum(0).squeeze()
nz = S_ != 0
c_[:, nz] /= S_[nz]
c_ = 1 - 2 * c_
c_[:, ~nz] = 0
dd = (hip_cv[0] - c_).abs()
print("HIP fused volume vs reference fusion formula on the HIP single-frame volumes: max %.3e, frac>1e-4 %.2e" % (dd.max(), (dd > 1e-4).float().mean()))
bad = (dd > 1e-4).any(0)
print("pixels involved:", int(bad.sum()), " S there (min / median / max):",
      (float(S_[bad].min()), float(S_[bad].median()), float(S_[bad].max())) if bad.any() else None)
dl = (ocv[0] - c_).abs()
print("oracle(this host) fused volume vs the same formula on HIP volumes: max %.3e, frac>1e-4 %.2e" % (dl.max(), (dl > 1e-4).float().mean()))
