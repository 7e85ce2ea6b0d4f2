"""Per-kernel register / LDS / scratch figures of one csrc/*.hip source as hipcc compiles it for gfx950 (no GPU needed).

    python tools/kernel_resources.py conv_mfma.hip [--grep conv_mfma_kernel] [--count v_mfma,ds_read]

Prints one line per kernel: VGPRs (arch + acc), SGPRs, spills, scratch bytes, static LDS, and - with --count - how often the given
mnemonic prefixes occur in the kernel's body.  Used before a hardware session to check that a source change did not cost occupancy.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_asm(src, extra_flags=()):
    from monorec_amd import build as _build
    flags = list(dict(_build.SOURCES).get(os.path.basename(src), [])) + list(extra_flags)
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([_build._hipcc(), f"--offload-arch={_build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c", src,
                        "-o", os.path.join(d, "x.o")] + flags, check=True, cwd=d, capture_output=True)
        stem = os.path.basename(src).rsplit(".", 1)[0]
        return open(os.path.join(d, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def kernels(asm):
    meta = asm[asm.index("amdhsa.kernels"):]
    out = []
    for blk in meta.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                        sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


def body(asm, mangled):
    m = re.search(r"^" + re.escape(mangled) + r":(.*?)\.Lfunc_end", asm, re.S | re.M)
    return m.group(1) if m else ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--grep", default="")
    ap.add_argument("--count", default="")
    ap.add_argument("--flag", action="append", default=[])
    a = ap.parse_args()
    src = os.path.abspath(a.source) if os.path.exists(a.source) else os.path.join(ROOT, "monorec_amd", "csrc", a.source)
    asm = compile_asm(src, a.flag)
    demangle = {}
    try:
        names = [k["name"] for k in kernels(asm)]
        res = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
        demangle = dict(zip(names, res.stdout.splitlines()))
    except Exception:
        pass
    for k in kernels(asm):
        pretty = demangle.get(k["name"], k["name"])
        if a.grep and a.grep not in pretty:
            continue
        line = f"{pretty[:110]:110s} vgpr {k['vgpr']:>4} agpr {k['agpr']:>3} sgpr {k['sgpr']:>4} spill v{k['vspill']} s{k['sspill']} scratch {k['scratch']} lds {k['lds']}"
        if a.count:
            b = body(asm, k["name"])
            line += "  " + " ".join("%s=%d" % (c, len(re.findall(r"^\s+" + re.escape(c), b, re.M))) for c in a.count.split(","))
        print(line)


if __name__ == "__main__":
    main()
