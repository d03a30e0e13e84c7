"""Compiled clients of the drop-in boundary (VERDICT r01 "next round" item 3).

tests/cabi_client/ref_client.c is written against the reference's own faer-ffi/faer.h: struct layouts are
_Static_assert-ed against include/faer_hip.h, the `libfaer_v0_23_*` prototypes it calls are the reference's, and it
links against libfaer_hip.so.  tests/cabi_client/hpp_client.cpp does the same through the reference's C++ wrapper
faer-ffi/faer.hpp (v0_24 spelling + six-scalar dispatch tables => csrc/abi_compat.c).  Building needs the
reference tree (here: /root/reference); the binaries land in tests/_build/cabi_client/ (git-ignored, they travel to
the GPU box), where the -m gpu test runs them in `compute` mode."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FAER_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "_build", "cabi_client")
LIBDIR = os.path.join(ROOT, "faer-rs_amd")


def build_clients():
    """returns the two binaries; raises CalledProcessError with the compiler output on a layout / prototype mismatch"""
    ffi = os.path.join(REF, "faer-ffi")
    os.makedirs(os.path.join(OUT, "hpp"), exist_ok=True)
    rpath = "-Wl,-rpath,$ORIGIN/../../../faer-rs_amd"
    ref_bin = os.path.join(OUT, "ref_client")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", f"-I{ffi}", f"-I{os.path.join(ROOT, 'include')}",
                    os.path.join(ROOT, "tests", "cabi_client", "ref_client.c"), "-o", ref_bin, f"-L{LIBDIR}", "-lfaer_hip", "-lm",
                    rpath], check=True, capture_output=True, text=True)
    # faer.hpp spells libfaer_v0_24_* but faer.h declares libfaer_v0_23_* (header drift in the reference): the C++
    # client compiles against a GENERATED copy of faer.h with the v0_24 spelling (build directory only, never committed)
    hdr = open(os.path.join(ffi, "faer.h")).read().replace("libfaer_v0_23_", "libfaer_v0_24_")
    open(os.path.join(OUT, "hpp", "faer.h"), "w").write(hdr)
    for f in ("faer.hpp", "quad.hpp"):
        shutil.copyfile(os.path.join(ffi, f), os.path.join(OUT, "hpp", f))
    hpp_bin = os.path.join(OUT, "hpp_client")
    subprocess.run(["g++", "-std=c++20", "-O1", f"-I{os.path.join(OUT, 'hpp')}", os.path.join(ROOT, "tests", "cabi_client", "hpp_client.cpp"),
                    "-o", hpp_bin, f"-L{LIBDIR}", "-lfaer_hip", rpath], check=True, capture_output=True, text=True)
    shutil.rmtree(os.path.join(OUT, "hpp"))  # copies of reference sources do not stay around (nor travel)
    return ref_bin, hpp_bin


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "faer-ffi", "faer.h")), reason="reference tree not available")
def test_clients_compile_link_and_run_host_only():
    try:
        bins = build_clients()
    except subprocess.CalledProcessError as e:  # show the compiler's message (e.g. the failing _Static_assert)
        pytest.fail(e.stderr[-4000:])
    for b in bins:
        r = subprocess.run([b], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "ok (host-only)" in r.stdout


def test_abi_compat_is_up_to_date(tmp_path):
    """csrc/abi_compat.c is generated from include/faer_hip.h: a fresh generation (into a temporary file -- the tracked
    file is never rewritten by the test run) must be identical"""
    path = os.path.join(LIBDIR, "csrc", "abi_compat.c")
    fresh = str(tmp_path / "abi_compat.c")
    subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_abi_compat.py"), fresh], check=True, capture_output=True)
    assert open(fresh).read() == open(path).read()


def test_v024_aliases_and_stubs_are_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libfaer_hip.so")], capture_output=True, text=True, check=True).stdout
    syms = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    v23 = {s for s in syms if s.startswith("libfaer_v0_23_") and s.endswith(("_f32", "_f64"))}
    assert len(v23) > 200
    for s in v23:
        assert s.replace("v0_23", "v0_24") in syms
        base = s.rsplit("_", 1)[0]
        for suf in ("fx128", "c32", "c64", "cx128"):
            assert f"{base}_{suf}" in syms and f"{base.replace('v0_23', 'v0_24')}_{suf}" in syms


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_client", "hpp_client"])
def test_clients_compute_on_the_gpu(name):
    b = os.path.join(OUT, name)
    if not os.path.exists(b):
        pytest.skip("client binaries are built where the reference tree is available (__graft_entry__.build())")
    r = subprocess.run([b, "compute"], capture_output=True, text=True, timeout=300)
    if r.returncode == 2 and "no gfx950 device" in (r.stdout + r.stderr):
        pytest.skip("no gfx950 device visible to the client")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok (compute)" in r.stdout
