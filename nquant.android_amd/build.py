"""Builds libnquant_hip.so (gfx950) in-tree with hipcc.  No torch, no JIT cache: the .so travels with the repo
snapshot to the GPU box.  -ffp-contract=off / no fast-math: parity with the reference arithmetic is bit for bit."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnquant_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# experiments: NQ_BUILD_TAG=x builds libnquant_hip.x.so from objects csrc/*.x.o with the extra defines NQ_BUILD_DEFS ("-DA=1 -DB"); the
# host mirror loads it when NQ_LIB points at it (host.py library_path)
TAG = os.environ.get("NQ_BUILD_TAG", "")
if TAG:
    LIB = os.path.join(HERE, "libnquant_hip.%s.so" % TAG)
EXTRA_DEFS = os.environ.get("NQ_BUILD_DEFS", "").split()
COMMON = (["-g"] if os.environ.get("NQ_BUILD_DEBUG") else []) + ["-O3", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
INC = os.path.join("..", "..", "include")
# object -> (kind, sources it is rebuilt for)
UNITS = {
    "nq_kernels.hip": ("device", ["nq_kernels.hip", "nq_device.h", "nq_kernels.h", "nq_dither.inc", "nq_palette.inc", "nq_merge.inc", "nq_lists.inc",
                                  os.path.join(INC, "nq_blue_noise_64x64.inc")]),
    "nq_dither_fast.hip": ("device", ["nq_dither_fast.hip", "nq_device.h", "nq_kernels.h", os.path.join(INC, "nq_blue_noise_64x64.inc")]),
    "nq_abi.cpp": ("host", ["nq_abi.cpp", "nq_kernels.h", os.path.join(INC, "nquant_abi.h")]),
}


STAMP = os.path.join(CSRC, ".build_flags" + ("." + TAG if TAG else ""))


def _flags():
    """Everything outside the sources that changes the objects: the timing-experiment switches must never leak into a later normal build."""
    return "debug=%s knockout=%s hipcc=%s common=%s defs=%s" % (bool(os.environ.get("NQ_BUILD_DEBUG")), bool(os.environ.get("NQ_BUILD_KNOCKOUT")),
                                                             HIPCC, " ".join(COMMON), " ".join(EXTRA_DEFS))


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    """Compiles the translation units whose sources changed (in parallel) and links libnquant_hip.so."""
    procs, objs = [], []
    me = os.path.abspath(__file__)
    flags = _flags()
    try:
        same_flags = open(STAMP).read() == flags
    except OSError:
        same_flags = False            # objects of unknown origin (if any): rebuild
    if not same_flags:
        force = True
    for src, (kind, deps) in UNITS.items():
        o = os.path.join(CSRC, src + (".%s.o" % TAG if TAG else ".o"))
        objs.append(o)
        if not force and not _newer(o, [os.path.join(CSRC, d) for d in deps] + [me]):
            continue
        if kind == "device":
            cmd = [HIPCC, "--offload-arch=gfx950", "-x", "hip"] + COMMON + EXTRA_DEFS + ["-c", os.path.join(CSRC, src), "-o", o]
            if os.environ.get("NQ_BUILD_KNOCKOUT"):     # timing experiments (tools/knockout.sh): stages can be left out at run time
                cmd.insert(-4, "-DNQ_FAST_KNOCKOUT")
        else:
            cmd = [HIPCC, "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + COMMON + EXTRA_DEFS + ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if procs:
        with open(STAMP, "w") as f:
            f.write(flags)
    if procs or force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
