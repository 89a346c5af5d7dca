"""Compile csrc/nq_kernels.hip with --save-temps into build/isa and print the register / scratch / occupancy summary and the
instruction mix of the kernels whose mangled name contains the given substring (default: gilbert_fast).
Usage: python tools/isa_stats.py [substring] [--nobuild]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "isa")
CSRC = os.path.join(ROOT, "nquant.android_amd", "csrc")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = args[0] if args else "gilbert_fast"
    unit = "nq_dither_fast" if "fast" in pat else "nq_kernels"
    SRC = os.path.join(CSRC, unit + ".hip")
    ASM = os.path.join(OUT, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    if "--nobuild" not in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-Wno-unused-function", "--save-temps", "--cuda-device-only", "-c", SRC, "-o", os.path.join(OUT, "dev.o")], cwd=OUT)
    text = open(ASM).read().split("\n")
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):", text[i])
        if m and pat in m.group(1):
            name = m.group(1)
            j = i + 1
            ops = collections.Counter()
            meta = {}
            while j < len(text) and "-- Begin function" not in text[j]:
                mm = re.match(r"^\s+([a-z][a-z0-9_]+)", text[j])
                if mm and not text[j].strip().startswith("."):
                    ops[mm.group(1)] += 1
                mk = re.match(r"^\s*; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize): (\S+)", text[j]) or \
                    re.match(r"^\s*; (codeLenInByte) = (\S+)", text[j])
                if mk:
                    meta[mk.group(1)] = mk.group(2)
                j += 1
            if meta:
                tot = sum(ops.values())
                cls = collections.Counter()
                for k, v in ops.items():
                    c = "f64" if "f64" in k else k.split("_")[0]
                    cls[c] += v
                print(name[:100])
                print("  ", meta)
                print("   instrs %d: %s" % (tot, dict(cls.most_common(8))))
                print("   top:", ", ".join("%s %d" % kv for kv in ops.most_common(28)))
            i = j
        else:
            i += 1


main()
