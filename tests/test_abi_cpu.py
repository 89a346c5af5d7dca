"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol include/nquant_abi.h
declares, and refuses to compute without a HIP device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import HAS_GPU, ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "nquant_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nq_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nq):
    L = nq.load_library()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libnquant_hip.so does not export %s" % name
    assert sorted(nq.abi_symbols()) == declared
    assert L.nq_abi_version() == 1


def test_library_does_not_link_the_oracle(nq):
    import subprocess
    out = subprocess.run(["ldd", nq.library_path()], capture_output=True, text=True).stdout
    assert "nq_oracle" not in out
    syms = subprocess.run(["nm", "-D", nq.library_path()], capture_output=True, text=True).stdout
    assert "nqo_" not in syms


def test_only_tests_smoke_and_the_cpu_baseline_touch_the_oracle():
    """oracle/ is test infrastructure: no file of the package, of tools/ or of include/ may import, load or name it; bench.py may
    only inside cpu_baseline(), __graft_entry__ only in build() (compiling the checker) and smoke()."""
    pat = re.compile(r"oracle_lib|libnq_oracle|nqo_[a-z]|import\s+oracle|from\s+oracle")
    offenders = []
    for top in ("nquant.android_amd", "nquant", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".inc", ".cpp", ".c", ".sh", ".java")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if pat.search(text):
                    offenders.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert offenders == [], offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench[bench.index("def cpu_baseline("):]
    body = body[:body.index("\ndef ", 10)]
    assert "oracle_lib" in body and "oracle_lib" not in bench.replace(body, "")


@pytest.mark.skipif(HAS_GPU, reason="checks the no-device error path")
def test_no_cpu_fallback(nq):
    with pytest.raises(nq.NqError) as e:
        nq.PnnLABQuantizer(np.zeros((8, 8), np.int32))
    assert e.value.status == -5 and "no CPU fallback" in str(e.value)


def test_host_mirror_argument_checks(nq):
    with pytest.raises(TypeError):
        nq.PnnQuantizer(np.zeros((4, 4), np.float32))


def test_params_struct_layout_matches_oracle(nq, oracle):
    import ctypes as C
    assert C.sizeof(nq.Params) == C.sizeof(oracle.Params) == 96
    assert [f for f, _ in nq.Params._fields_] == [f for f, _ in oracle.Params._fields_]


def test_synth_is_deterministic(nq):
    from nquant.android_amd import synth
    a = synth.uniform_rgb(64, 64, 1)
    assert a.dtype == np.int32 and a.shape == (64, 64)
    assert (a.view(np.uint32) >> 24 == 255).all()
    assert int(a.view(np.uint32).astype(np.uint64).sum()) == int(synth.uniform_rgb(64, 64, 1).view(np.uint32).astype(np.uint64).sum())
    g = synth.gradient_noise(96, 64, 3)
    assert g.shape == (64, 96) and (g.view(np.uint32) >> 24 == 255).all()
    t = synth.with_alpha(g, 3)
    al = t.view(np.uint32) >> 24
    assert (al == 0).any() and ((al > 15) & (al < 0xE0)).any()


def _java_natives(path):
    """{name: number of parameters} of the `native` methods a Java host class declares."""
    src = open(path).read()
    out = {}
    for m in re.finditer(r"\bnative\s+[\w\[\]\.]+\s+(\w+)\s*\(([^)]*)\)", src, re.S):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params)
    return out


def test_jni_shim_matches_the_header_and_the_java_host_class():
    """SURVEY 8f row 1 cannot run here (no JDK): what CAN be held is that the shim stays in step with the ABI it wraps.  (1) nquant_jni.c
    passes `gcc -fsyntax-only -Wall -Werror` against include/nquant_abi.h and a stub of the few JNI declarations it uses
    (tests/c/jni_syntax/jni.h -- not a JDK header, never linked): a changed nq_* signature breaks this test; (2) every `native` method of
    the Java host class has its Java_com_android_nQuant_PnnQuantizer_<name> function with the same number of parameters (+ env, class),
    and the reverse; (3) the Java class keeps the reference's public surface (constructor from a file name, convert(int, boolean)
    throws Exception, hasAlpha(): NQ/PnnQuantizer.java:35,409,458)."""
    import subprocess
    shim = os.path.join(ROOT, "nquant.android_amd", "jni", "nquant_jni.c")
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "c", "jni_syntax"),
                        "-I", os.path.join(ROOT, "include"), shim], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    java = os.path.join(ROOT, "nquant.android_amd", "java", "com", "android", "nQuant", "PnnQuantizer.java")
    natives = _java_natives(java)
    csrc = re.sub(r"/\*.*?\*/", "", open(shim).read(), flags=re.S)
    cfuncs = {m.group(1): len([p for p in m.group(2).split(",") if p.strip()])
              for m in re.finditer(r"Java_com_android_nQuant_PnnQuantizer_(\w+)\s*\(([^)]*)\)", csrc, re.S)}
    assert sorted(natives) == sorted(cfuncs) and len(natives) >= 5
    for name, n in natives.items():
        assert cfuncs[name] == n + 2, (name, n, cfuncs[name])
    jsrc = open(java).read()
    assert re.search(r"public\s+PnnQuantizer\s*\(\s*String\s+\w+\s*\)", jsrc)
    assert re.search(r"public\s+Bitmap\s+convert\s*\(\s*int\s+nMaxColors\s*,\s*boolean\s+dither\s*\)\s*throws\s+Exception", jsrc)
    assert re.search(r"public\s+boolean\s+hasAlpha\s*\(\s*\)", jsrc)
    lab = open(os.path.join(os.path.dirname(java), "PnnLABQuantizer.java")).read()
    assert re.search(r"class\s+PnnLABQuantizer\s+extends\s+PnnQuantizer", lab) and re.search(r"public\s+PnnLABQuantizer\s*\(\s*String\s+\w+\s*\)", lab)
