"""Build libmonorec_hip.so (the C-ABI of include/monorec_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmonorec_hip.so")
ARCH = "gfx950"

# (source, extra flags). cost_volume.hip reproduces the CPU reference operation by operation, so the
# compiler must not contract a*b+c on its own (explicit fmaf marks the places the reference fuses).
SOURCES = [
    ("conv_mfma.hip", []),
    # -fno-slp-vectorize: the marching cost-volume kernel lives on DPP-fused adds; the SLP pass pairs them into v_pk_add_f32,
    # which cannot carry a DPP shift (measured on the ISA: +20 % VALU instructions, 170 instead of 123 VGPRs)
    ("cost_volume.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("pointcloud.hip", ["-ffp-contract=off"]),
    ("preprocess.hip", ["-ffp-contract=off"]),
    ("eltwise.hip", []),
    ("heads.hip", []),
    ("conv_wino.hip", []),
    ("convt_wino.hip", []),
    ("conv1d_wino.hip", []),
    ("conv_b8.hip", []),
    # (packed f32 VALU beside MFMAs is priced as an anti-lever in MI355X_MICROARCH.md; the flag on ALL MFMA kernels was measured in
    # tools/sessions/r04_s27.sh: c2 739 vs 740, c3 654 vs 655, configs[4] bf16 333 vs 330 keyframes/s - noise, so only this kernel has it)
    ("conv_wino44.hip", ["-fno-slp-vectorize"]),
    ("conv_wino44s.hip", ["-fno-slp-vectorize"]),      # F(4x4,3x3) with the positions split over two waves: two workgroups per CU (round 5)
    ("conv_wino44w.hip", ["-fno-slp-vectorize"]),      # F(4x4,3x3) with one wave per SIMD: both cout blocks per wave, quad pipeline in registers (round 6)
]
# conv_wino44.hip (F(4x4,3x3)): out of the product library for most of round 4 (at c2 it moved keyframes/s by nothing),
# back in once its c3 / configs[4] tables were measured (tools/sessions/r04_s18.sh): 12-15 % ahead of the best F(2x2,3x3) variant on every
# full- and half-resolution layer of those shapes (c3 mask.enc0.* 1682 -> 1469 us each; 0.63 ms of a 13.9 ms batch).  The F(2,7)
# instantiations of conv1d_wino.hip, which no table entry ever selected, stay in the DIAGNOSTIC library only.
DIAGNOSTIC_ONLY_SOURCES = []


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _run_all(cmds, verbose=False):
    """The stale translation units side by side (conv_mfma.hip alone takes ~3 minutes: one after the other a full rebuild was 10+)."""
    if not cmds:
        return
    from concurrent.futures import ThreadPoolExecutor

    def one(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(cmds), (os.cpu_count() or 2)))) as pool:
        list(pool.map(one, cmds))


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB_PATH) and not _stale(
            LIB_PATH, [os.path.join(CSRC, s) for s, _ in SOURCES] + [os.path.join(CSRC, "conv_layout.h"), os.path.join(CSRC, "cooktoom_1d.h"), __file__,
                                                                       os.path.join(HERE, "..", "include", "monorec_hip.h")]):
        return LIB_PATH                     # prebuilt library travels with the snapshot; nothing to do
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "conv_layout.h"), os.path.join(CSRC, "cooktoom_1d.h"), os.path.join(HERE, "..", "include", "monorec_hip.h"), __file__]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, cmds = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmds.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o] + extra)
    _run_all(cmds, verbose)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


TIMELINE_LIB_PATH = os.path.join(HERE, "libmonorec_hip_timeline.so")


# sources whose diagnostic switches are compiled in only for the diagnostic library: (source, define)
DIAGNOSTIC = {"conv_mfma.hip": "-DMR_CONV_TIMELINE",     # per-workgroup timestamps / ablation bits (MR_CONV_DBG; ~3 % slower even when off)
              "cost_volume.hip": "-DMR_TUNING_ENV",      # MR_CV_MARCH_TY / MR_CV_MARCH_DP / MR_CV_NO_KF_PREPASS (tools/bench_cv.py sweeps)
              "heads.hip": "-DMR_TUNING_ENV",            # MR_HEADS_QUAD_MIN (tools/bench_heads.py)
              "conv1d_wino.hip": "-DMR_DIAGNOSTIC_FORMS",   # the F(2,7) instantiations
              "eltwise.hip": "-DMR_DIAGNOSTIC_FORMS",       # exports mr_diagnostic_forms(): how _lib.load() tells the two builds apart
              "conv_b8.hip": "-DMR_B8_ABLATE",              # MR_B8_DBG ablation bits (tools/bench_b8.py)
              "conv_wino44.hip": "-DMR_W44_ABLATE"}         # MR_W44_DBG ablation bits (tools/sessions/r04_s23.sh)


def build_timeline(verbose=False):
    """Diagnostic variant of the library (tools/wg_timeline.py, the MR_* tuning sweeps): the sources in DIAGNOSTIC recompiled with
    their switches; the product library reads no environment variable on any launch path.  Select it with MR_HIP_LIBRARY."""
    build(verbose=verbose)
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    flags = dict(SOURCES)
    objs, cmds = [], []
    hdrs = [os.path.join(CSRC, "conv_layout.h"), os.path.join(CSRC, "cooktoom_1d.h"), os.path.join(HERE, "..", "include", "monorec_hip.h")]
    for src, _ in SOURCES:
        if src not in DIAGNOSTIC:
            objs.append(os.path.join(objdir, src.replace(".hip", ".o")))
            continue
        o = os.path.join(objdir, src.replace(".hip", "_diag.o"))
        s = os.path.join(CSRC, src)
        if _stale(o, [s] + hdrs):
            cmds.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", DIAGNOSTIC[src], "-DMR_DIAGNOSTIC_LIBRARY", "-c", s, "-o", o] + flags[src])
        objs.append(o)
    _run_all(cmds, verbose)
    for src, extra in DIAGNOSTIC_ONLY_SOURCES:
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        s = os.path.join(CSRC, src)
        if _stale(o, [s, os.path.join(CSRC, "conv_layout.h"), os.path.join(CSRC, "cooktoom_1d.h"), os.path.join(HERE, "..", "include", "monorec_hip.h")]):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-DMR_DIAGNOSTIC_LIBRARY", "-c", s, "-o", o] + extra
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if _stale(TIMELINE_LIB_PATH, objs):
        subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", TIMELINE_LIB_PATH] + objs, check=True)
    return TIMELINE_LIB_PATH


if __name__ == "__main__":
    if "--timeline" in sys.argv:
        print(build_timeline(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
