"""Instruction budget of gilbert_fast_kernel<2, 1> by block ("dither kernel: where the instructions go").
Compiles csrc/nq_dither_fast.hip with -DNQ_FAST_MARKS (asm comments between the blocks of one pixel step; the markers carry no
instruction and the marked build is never shipped) into build/isa_marks and counts the instructions of the LAB kernel between consecutive
markers in layout order.  The step's blocks are laid out in program order, so the counts are the STATIC instructions on the path of a step;
what a wavefront executes differs where a block holds a loop (the rolled candidate loops beyond eight entries, the limiter's three turns)
or exec-masked branches that no lane takes (the f64 fallbacks).  Of the five accumulation bodies and the five queue pushes one runs per step.
Usage: python tools/fast_budget.py [--nobuild]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "isa_marks")
SRC = os.path.join(ROOT, "nquant.android_amd", "csrc", "nq_dither_fast.hip")
ASM = os.path.join(OUT, "nq_dither_fast-hip-amdgcn-amd-amdhsa-gfx950.s")


def klass(op):
    if op.startswith(("v_pk_", )): return "valu_pk_f32"
    if "f64" in op: return "valu_f64"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")): return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait/nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


def main():
    if "--nobuild" not in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-Wno-unused-function", "-DNQ_FAST_MARKS", "--save-temps", "--cuda-device-only", "-c", SRC,
                               "-o", os.path.join(OUT, "dev.o")], cwd=OUT)
    lines = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2nq19gilbert_fast_kernelILi2ELi1E", l))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    segs, cur, name = [], collections.Counter(), "prologue"
    for l in lines[start:end]:
        m = re.match(r"^\s*; (MARK \w+|window offset \d|push \d)", l)
        if m:
            segs.append((name, cur))
            cur, name = collections.Counter(), m.group(1).replace("MARK ", "")
            continue
        mm = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if mm and not l.strip().startswith((".", ";")):
            cur[klass(mm.group(1))] += 1
    segs.append((name, cur))
    # a segment is named after the marker that OPENS it; report it as "what lies between this marker and the next"
    label = {"step_begin": "pixel -> floats, (one of five) accumulation bodies up to the first `window offset`",
             "window offset 4": "accumulation body (window offset 3)", "window offset 3": "accumulation body (window offset 2)",
             "window offset 2": "accumulation body (window offset 1)", "window offset 1": "accumulation body (window offset 0)",
             "window offset 0": "clamp + pack c2", "accumulated": "ditherPixel colour (fast_dither_color: Y_Diff tests, kappa, BlueNoise.diffuse)",
             "dither_color": "cell index + gather of the 32-byte list record", "lists_fetched": "closestColorIndex: 8 straight-line candidates + rolled tail + f64 arbiter",
             "closest_done": "java.util.Random draw + modulo + pick", "picked": "nearestColorIndex: float32 Lab + 8 straight-line candidates + tail + exact fallback call",
             "nearest_done": "error = adjusted - palette colour, blue-noise gate", "error_formed": "limiter loop body (runs three times per step): tanh / illusion / division",
             "limited": "queue push (case 3 incl. nothing else)", "push 3": "queue push", "push 2": "queue push", "push 1": "queue push",
             "push 0": "queue push case 0 + the window move (once per five steps) + index staging", "step_end": "loop control, next-pixel prefetch (laid out ahead of step_begin)"}
    order = ["valu", "valu_pk_f32", "valu_f64", "salu", "branch", "lds", "vmem", "wait/nop"]
    print("%-26s %s   what" % ("segment (opening marker)", " ".join("%9s" % o for o in order)))
    first = True
    for name, c in segs:
        if name == "prologue" and first:
            first = False
            continue
        if name in ("prologue",):
            continue
        print("%-26s %s   %s" % (name, " ".join("%9d" % c.get(o, 0) for o in order), label.get(name, "")))
        if name == "step_end" and not first and segs.index((name, c)) > 3:
            break


main()
