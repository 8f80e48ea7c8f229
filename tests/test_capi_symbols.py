"""The C-ABI library loads and exports every symbol include/smhip.h declares (no GPU)."""
import ctypes
import os
import re

from stereomatch_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "smhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smhip_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_symbol():
    assert os.path.exists(capi.LIB_PATH), "build libsmhip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    capi.load()


def test_match_outside_needs_no_gpu():
    L = capi.load()
    p = capi.match_params(384, 288, 4, 16, match_fn=2, match_max=1000)
    assert L.smhip_match_outside(ctypes.byref(p)) == 4 * 255 * 255      # StcRawCosts.cpp:365-367
    p = capi.match_params(384, 288, 4, 16, match_fn=1, match_max=20)
    assert L.smhip_match_outside(ctypes.byref(p)) == 20


def test_header_is_plain_c99():
    """include/smhip.h is what a cgo / JNI / ctypes binding reads: it has to be C, not C++."""
    import subprocess
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-x", "c", os.path.join(ROOT, "include", "smhip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_demo_links(tmp_path):
    """examples/c_abi_demo.c (a C99 caller) compiles and links against the library without warnings."""
    import subprocess
    lib_dir = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", lib_dir, "-lsmhip",
                        "-Wl,-rpath," + lib_dir, "-o", str(tmp_path / "c_abi_demo")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
