"""Loader for tests/cpu_harness/libharness.so: the device headers compiled for
the host with bounds checks.  Test-only."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "cpu_harness")
_SO = os.path.join(_DIR, "libharness.so")
_CSRC = os.path.join(os.path.dirname(_HERE), "bulletproofs_amd", "csrc")


def build():
    srcs = [os.path.join(_DIR, "harness.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", _SO,
                               os.path.join(_DIR, "harness.cpp")])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib
