"""The drop-in boundary: libppals.so loads on a box without a GPU and exports every entry point
include/ppals.h declares (no compute calls here); creating a context without a device fails loudly
with PPALS_ERR_NO_DEVICE instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pairwise-perturbation_amd", "lib", "libppals.so")
HDR = os.path.join(ROOT, "include", "ppals.h")


def declared_functions():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ppals_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: run __graft_entry__.build() / make -C pairwise-perturbation_amd")
    return C.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_lists_the_same_symbols():
    import ppals
    assert sorted(ppals.EXPORTS) == declared_functions()


def test_no_device_is_an_error_not_a_fallback(lib):
    if os.path.exists("/dev/kfd"):   # (not through torch: its own HIP runtime must not be mixed in)
        pytest.skip("a GPU may be present")
    ctx = C.c_void_p()
    rc = lib.ppals_ctx_create(C.byref(ctx), 0)
    lib.ppals_last_error.restype = C.c_char_p
    assert rc == -1 and not ctx.value          # PPALS_ERR_NO_DEVICE
    assert b"no HIP device" in lib.ppals_last_error() or b"HIP" in lib.ppals_last_error()
