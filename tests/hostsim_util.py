"""Loads the ppals ctypes binding against tests/hostsim/build/libppals_hostsim.so (TEST
INFRASTRUCTURE: the product's engine + C-ABI sources over a host stand-in for the device ops)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_mod = None


def load():
    global _mod
    if _mod is None:
        libpath = os.environ.get("PPALS_HOSTSIM_LIB")  # tests/test_sanitizers.py: the ASan build
        if not libpath:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
            libpath = os.path.join(ROOT, "tests", "hostsim", "build", "libppals_hostsim.so")
        path = os.path.join(ROOT, "pairwise-perturbation_amd", "ppals", "__init__.py")
        spec = importlib.util.spec_from_file_location("ppals_hostsim", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ppals_hostsim"] = mod  # so objects can find the binding they came from
        spec.loader.exec_module(mod)
        mod._LIBPATH = libpath
        assert b"hostsim" in mod.lib().ppals_version()
        _mod = mod
    return _mod
