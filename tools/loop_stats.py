#!/usr/bin/env python3
"""Instruction counts per ray step of the lean trace kernels, from the compiler's assembly.

usage: python tools/loop_stats.py [trace_kernels.s]   (default: compile artist_amd/csrc/trace_kernels.hip to /tmp first)
A ray step of the ring loop starts at its `s_waitcnt vmcnt(depth - 1)`; the steps of a round are averaged.
"""
import collections
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
args = [x for x in sys.argv[1:] if not x.startswith("--")]
if args:
    asm = pathlib.Path(args[0])
else:
    asm = pathlib.Path("/tmp/trace_kernels_stats.s")
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize".split()
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", str(asm), "trace_kernels.hip"],
                   cwd=ROOT / "artist_amd" / "csrc", check=True, stderr=subprocess.DEVNULL)
lines = asm.read_text().split("\n")


SRC = (ROOT / "artist_amd" / "csrc" / "trace_kernels.hip").read_text()


def ring_depth(macro):
    """The ring depth of the shipped source: tools/strip_variants.py left it as a comment, "// ring depth = 8: ..." /
    "// ring depth bwd = 2: ..." (the instrumented copy under tools/diag/ still has the #define)."""
    words = macro[4:].lower().replace("_", " ")
    m = re.search(r"//\s*" + words + r"\s*=\s*(\d+)", SRC) or re.search(r"#define\s+" + macro + r"\s+(\d+)", SRC)
    return int(m.group(1)) if m else 8


def stats(name, pat, depth):
    s = [i for i, l in enumerate(lines) if l.startswith(pat)][0]
    e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    # a ring step: the wait for its slot, then the (inline asm) copy of the slot into the ray's own registers
    w = [i for i in range(s, e) if f"s_waitcnt vmcnt({depth - 1})" in lines[i] and "#ASMSTART" in lines[i + 1] and "v_mov_b32" in lines[i + 2]]
    span = None
    for k in range(len(w) - depth + 1):
        d = [w[k + i + 1] - w[k + i] for i in range(depth - 1)]
        if max(d) - min(d) < 40:
            span = (w[k], w[k + depth - 1] + sum(d) // (depth - 1))
            break
    if span is None:
        print(name, "ring loop not found")
        return
    body = [l for l in lines[span[0]:span[1]] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    ins = [l.split()[0] for l in body]
    # "slow class" of DESIGN.md section 4.0: a VALU instruction with a scalar-register or literal operand, or one of the
    # opcodes that take a second pass through the pipe whatever their operands
    slow_ops = ("v_trunc", "v_cvt", "v_med3", "v_max", "v_min", "v_cmp", "v_mad_u32", "v_lshl_add", "v_or3", "v_pk_", "v_cndmask",
                "v_rcp", "v_rsq", "v_sqrt", "v_readlane", "v_readfirstlane")
    n_slow_op = n_slow_operand = 0
    for l in body:
        op = l.split()[0]
        if not op.startswith("v_"):
            continue
        operands = l.split(";")[0].split(None, 1)[1] if len(l.split(";")[0].split(None, 1)) > 1 else ""
        srcs = [x.strip() for x in operands.split(",")][1:]
        if op.startswith(slow_ops):
            n_slow_op += 1
        elif any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|0x|-?\d+\.\d|[0-9a-fx]+$)", x) and not re.match(r"^-?[0-4]$|^-?(0\.5|1\.0|2\.0|4\.0)$|^0$", x) for x in srcs):
            n_slow_operand += 1
    c = collections.Counter()
    for k in ins:
        if k.startswith("v_"):
            c["valu"] += 1
        elif k.startswith(("s_waitcnt", "s_nop")):
            c["wait"] += 1
        elif k.startswith("s_"):
            c["salu"] += 1
        elif k.startswith("ds_"):
            c["lds"] += 1
        elif k.startswith(("global_", "scratch_", "buffer_")):
            c["vmem"] += 1
    meta = {}
    for i, l in enumerate(lines):
        if l.strip().startswith(".name:") and pat.rstrip(":") in l:
            for j in range(i - 70, i + 30):
                m = re.match(r"\s+\.(vgpr_count|private_segment_fixed_size|sgpr_count):\s+(\d+)", lines[j])
                if m:
                    meta[m.group(1)] = int(m.group(2))
            break
    if "--histogram" in sys.argv:
        hist = collections.Counter(ins)
        print(f"  opcodes per ray, {name}:")
        for k, v in hist.most_common():
            print(f"    {k:28s} {v / depth:6.2f}")
    print(f"{name} (ring of {depth}): per ray " + ", ".join(f"{k} {v / depth:.1f}" for k, v in sorted(c.items())) +
          f"; slow-class opcodes {n_slow_op / depth:.1f}, scalar / literal operands {n_slow_operand / depth:.1f}; v_mov {sum(k.startswith('v_mov') for k in ins) / depth:.1f}, scratch in loop {sum(k.startswith('scratch') for k in ins)}; {meta}")


stats("forward lean (interleaved)", "_ZN3art20trace_fwd_lds_kernelILb1ELb0ELb0ELi1EEEv", ring_depth("ART_RING_DEPTH"))
stats("backward lean (interleaved)", "_ZN3art20trace_bwd_lds_kernelILb1ELb0ELb0ELb0ELb1EEEv", ring_depth("ART_RING_DEPTH_BWD"))
