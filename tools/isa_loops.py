#!/usr/bin/env python
"""Hot-loop census of every MFMA kernel as hipcc compiles it for gfx950 (no GPU needed): for each kernel of a csrc/*.hip source, every
INNERMOST loop that contains matrix instructions, with the instruction mix of its body -

    python tools/isa_loops.py [conv_mfma.hip conv_wino44.hip ...] [--grep conv_mfma_kernel<1, 1] [--out profiles/r06_isa_loops.json]

Why the mix matters (round 6, tools/probes/mfma_rates.hip + profiles/r06_c2_sweep_ablation.json): `v_mfma_f32_16x16x4_f32` runs on the SIMD's
fp32 vector ALUs, so every VALU instruction in the loop takes its issue time out of the matrix stream (32.0 cycles per MFMA back to back,
43-55 with one v_add_u32 + two SALU per MFMA) - `valu_per_mfma` is the figure to drive down for the fp32 kernels; `pattern` shows whether
operand reads of the next k-steps are in flight while the MFMAs of this one issue (L = LDS read, w = s_waitcnt lgkmcnt, M = MFMA, V = VALU,
G = global / buffer access; SALU and branches are left out of the string).
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.kernel_resources import compile_asm, kernels, body  # noqa: E402

MFMA_SOURCES = ["conv_mfma.hip", "conv_wino.hip", "conv_wino44.hip", "conv_wino44s.hip", "conv_wino44w.hip", "conv1d_wino.hip", "convt_wino.hip", "conv_b8.hip"]


def classify(ins):
    if ins.startswith("v_mfma") or ins.startswith("v_smfmac"):
        return "M"
    if ins.startswith("ds_read") or ins.startswith("ds_load"):
        return "L"
    if ins.startswith("ds_write") or ins.startswith("ds_store"):
        return "S"
    if ins.startswith("s_waitcnt"):
        return "w"
    if ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "G"
    if ins.startswith("v_"):
        return "V"
    if ins.startswith(("s_cbranch", "s_branch")):
        return "b"
    if ins.startswith("s_barrier"):
        return "B"
    if ins.startswith("s_nop"):
        return "n"
    if ins.startswith("s_"):
        return "s"
    return "?"


def loops_of(text):
    """(label, first line, last line) of every loop = a conditional / unconditional branch back to a label defined above it."""
    lines = text.split("\n")
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            out.append((m.group(1), label_at[m.group(1)], i))
    return lines, out


def census(asm_body):
    lines, loops = loops_of(asm_body)
    res = []
    for lab, a, b in loops:
        if any(a <= a2 and b2 <= b and (a2, b2) != (a, b) and any(re.match(r"^\s+v_s?mfma", l) for l in lines[a2:b2 + 1]) for _, a2, b2 in loops):
            continue                                   # not innermost among the loops that hold MFMAs
        seq = []
        for ln in lines[a:b + 1]:
            m = re.match(r"^\s+([a-z_0-9]+)", ln)
            if m:
                seq.append((classify(m.group(1)), m.group(1), ln))
        n = {k: sum(1 for c, _, _ in seq if c == k) for k in "MLSwGVsbBn"}
        if not n["M"]:
            continue
        pat = "".join(c for c, _, _ in seq if c in "MLwVGSB")
        pat = re.sub(r"(.)\1*", lambda m_: m_.group(1) + (str(len(m_.group(0))) if len(m_.group(0)) > 1 else ""), pat)
        mf = sorted({i for c, i, _ in seq if c == "M"})
        res.append(dict(label=lab, mfma=n["M"], mfma_ops=mf, lds_reads=n["L"], lds_writes=n["S"], waits=n["w"], valu=n["V"], salu=n["s"], vmem=n["G"], branches=n["b"], barriers=n["B"],
                        nops=n["n"], valu_per_mfma=round(n["V"] / n["M"], 3), lds_reads_per_mfma=round(n["L"] / n["M"], 3), pattern=pat))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="*", default=MFMA_SOURCES)
    ap.add_argument("--grep", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--flag", action="append", default=[])
    a = ap.parse_args()
    report = {}
    for src in a.sources or MFMA_SOURCES:
        path = src if os.path.exists(src) else os.path.join(ROOT, "monorec_amd", "csrc", src)
        asm = compile_asm(path, a.flag)
        ks = kernels(asm)
        names = [k["name"] for k in ks]
        try:
            res = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
            pretty = dict(zip(names, res.stdout.splitlines()))
        except Exception:
            pretty = {}
        for k in ks:
            nm = pretty.get(k["name"], k["name"])
            if a.grep and a.grep not in nm:
                continue
            lp = census(body(asm, k["name"]))
            if not lp:
                continue
            report[nm] = dict(source=os.path.basename(path), vgpr=k["vgpr"], agpr=k["agpr"], sgpr=k["sgpr"], lds_static=k["lds"], loops=lp)
            hot = max(lp, key=lambda l: l["mfma"])
            print(f"{nm[:100]:100s} vgpr {k['vgpr']:>4}  loops {len(lp)}  hottest: {hot['mfma']} MFMA, {hot['lds_reads']} LDS reads, {hot['valu']} VALU "
                  f"({hot['valu_per_mfma']}/MFMA), {hot['salu']} SALU  {hot['pattern'][:90]}")
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"note": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": report}, f, indent=1)


if __name__ == "__main__":
    main()
