"""Builds libnquant_hip.so (gfx950) in-tree with hipcc.  No torch, no JIT cache: the .so travels with the repo
snapshot to the GPU box.  -ffp-contract=off / no fast-math: parity with the reference arithmetic is bit for bit."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnquant_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
DEVICE_SOURCES = ["nq_kernels.hip"]
HOST_SOURCES = ["nq_abi.cpp"]
DEPS = ["nq_device.h", "nq_kernels.h", "nq_dither.inc", "nq_palette.inc", "nq_merge.inc", "nq_lists.inc",
        os.path.join("..", "..", "include", "nquant_abi.h"), os.path.join("..", "..", "include", "nq_blue_noise_64x64.inc")]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    all_src = [os.path.join(CSRC, s) for s in DEVICE_SOURCES + HOST_SOURCES + DEPS] + [os.path.abspath(__file__)]
    if not force and not _newer(LIB, all_src):
        return LIB
    objs = []
    for s in DEVICE_SOURCES:
        o = os.path.join(CSRC, s + ".o")
        cmd = [HIPCC, "--offload-arch=gfx950", "-x", "hip"] + COMMON + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    for s in HOST_SOURCES:
        o = os.path.join(CSRC, s + ".o")
        cmd = [HIPCC, "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + COMMON + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
