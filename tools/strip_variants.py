#!/usr/bin/env python3
"""Resolve the build-time A/B, ablation and timeline variants of a kernel source to the SHIPPED configuration.

Rounds 1-3 grew ~90 ``#ifdef ART_*`` blocks in the trace kernels (diagnostic timelines, ablation builds that compute wrong results on
purpose to price a stage, A/B bodies).  The shipped translation units carry none of them: this script evaluates every conditional
on an ``ART_*`` macro with the product's values (flags undefined, sizes at their defaults), substitutes the size macros by their
numbers and drops the timeline marks.  The instrumented sources of round 3 stay under tools/diag/ for the measurement scripts
(tools/build_obj_variant.sh builds them with -D flags into a separate library; they are not part of the product build).
usage: python tools/strip_variants.py <in> <out>"""
import re
import sys

VALUES = {"ART_RING_DEPTH": 8, "ART_RING_DEPTH_BWD": 2, "ART_LEAN_FWD_THREADS": 1024, "ART_LEAN_BLOCK_FWD_THREADS": 768,
          "ART_LEAN_CYL_FWD_THREADS": 1024, "ART_LEAN_BWD_THREADS": None, "ART_LEAN_BLOCK_BWD_THREADS": None, "ART_CYL_BWD_THREADS": None,
          "ART_COM_UNROLL": 4, "ART_CROP_UNROLL": 4, "ART_CROP_TILE_Y": 32, "ART_CROP_TILE_ROWS": 64, "ART_CROP_ROW_UNROLL": 4}


def main(src, dst):
    lines = open(src).read().split("\n")
    # defaults of the size macros as written in the source: "#ifndef X / #define X v / #endif"
    for i, ln in enumerate(lines):
        m = re.match(r"#ifndef (ART_[A-Z_0-9]+)\b", ln)
        if m and m.group(1) in VALUES and i + 1 < len(lines):
            d = re.match(r"#define %s\s+(\S+)" % m.group(1), lines[i + 1])
            if d:
                VALUES[m.group(1)] = int(d.group(1))
    out, stack = [], []          # stack of [emitting_before, taken_already, emitting_now, is_art]

    def evaluate(expr):
        expr = re.sub(r"//.*", "", expr).strip()
        expr = re.sub(r"defined\s*\(\s*(ART_[A-Z_0-9]+)\s*\)", lambda m: "1" if VALUES.get(m.group(1)) is not None and m.group(1) in VALUES and False else "0", expr)
        for k, v in VALUES.items():
            if v is not None:
                expr = re.sub(r"\b%s\b" % k, str(v), expr)
        if re.search(r"[A-Za-z_]", expr):
            return None
        return bool(eval(expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ")))

    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        emitting = all(f[2] for f in stack)
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if m:
            kind, rest = m.group(1), m.group(2)
            if kind in ("ifdef", "ifndef"):
                name = rest.split()[0]
                if name.startswith("ART_"):
                    if name in VALUES and kind == "ifndef" and i + 1 < len(lines) and re.match(r"#define %s\b" % name, lines[i + 1].strip()):
                        # the default of a size macro ("#ifndef X / #define X v // why / ... / #endif"): the number is substituted
                        # below; what is kept is the comment
                        j = i + 1
                        comment = []
                        while not lines[j].strip().startswith("#endif"):
                            c = re.search(r"//(.*)", lines[j])
                            if c:
                                comment.append(c.group(1).strip())
                            j += 1
                        if emitting and comment:
                            out.append("// " + name[4:].lower().replace("_", " ") + " = " + str(VALUES[name]) + ": " + comment[0])
                            out.extend("// " + c for c in comment[1:])
                        i = j + 1
                        continue
                    defined = False                       # every ART_ flag is undefined in the product
                    take = (not defined) if kind == "ifndef" else defined
                    stack.append([emitting, take, take, True])
                    i += 1
                    continue
                stack.append([emitting, True, True, False])
                if emitting:
                    out.append(ln)
                i += 1
                continue
            if kind == "if":
                val = evaluate(rest) if "ART_" in rest else None
                if val is None:
                    stack.append([emitting, True, True, False])
                    if emitting:
                        out.append(ln)
                else:
                    stack.append([emitting, val, val, True])
                i += 1
                continue
            top = stack[-1]
            if kind == "elif":
                if top[3]:
                    val = evaluate(rest)
                    assert val is not None, ln
                    top[2] = (not top[1]) and val
                    top[1] = top[1] or top[2]
                elif all(f[2] for f in stack[:-1]):
                    out.append(ln)
                i += 1
                continue
            if kind == "else":
                if top[3]:
                    top[2] = not top[1]
                    top[1] = True
                elif all(f[2] for f in stack[:-1]):
                    out.append(ln)
                i += 1
                continue
            if kind == "endif":
                stack.pop()
                if not top[3] and all(f[2] for f in stack):
                    out.append(ln)
                i += 1
                continue
        if emitting:
            out.append(ln)
        i += 1
    assert not stack
    text = "\n".join(out)
    # timeline marks: the macro is empty in the product
    text = re.sub(r"^#define ART_TIMELINE\(k\)\s*\n", "", text, flags=re.M)
    text = re.sub(r"^[ \t]*ART_TIMELINE\(\d+\);[ \t]*\n", "", text, flags=re.M)
    text = re.sub(r"[ \t]*ART_TIMELINE\(\d+\);", "", text)
    for k, v in VALUES.items():
        if v is not None:
            text = re.sub(r"\b%s\b" % k, str(v), text)
    open(dst, "w").write(text)
    left = re.findall(r"^#\s*if.*ART_.*$", text, flags=re.M)
    print(f"{src} -> {dst}: {len(lines)} -> {text.count(chr(10)) + 1} lines; conditionals on ART_ left: {left}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
