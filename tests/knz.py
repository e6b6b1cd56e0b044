"""Test helper: imports the product package (directory name 'kanzi-go_amd' is not a Python identifier) and, for the
CPU-container tests, builds the same kernel sources against the HIP execution-model emulator (tests/emu)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "kanzi-go_amd")


def package():
    if "kanzi_go_amd" in sys.modules:
        return sys.modules["kanzi_go_amd"]
    spec = importlib.util.spec_from_file_location("kanzi_go_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kanzi_go_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def emu_library():
    """TEST INFRASTRUCTURE: the kernels compiled for the CPU emulator. Never used by bench.py / smoke()."""
    out = os.path.join(ROOT, "tests", "emu", "build", "libknz_gpu_emu.so")
    srcs = [os.path.join(PKG_DIR, "csrc", f) for f in os.listdir(os.path.join(PKG_DIR, "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "emu", "hip_emu.cpp"), os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h"),
             os.path.join(ROOT, "include", "knz_gpu.h")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++",
                               "-I", os.path.join(ROOT, "tests", "emu", "include"), "-Wno-unknown-pragmas",
                               os.path.join(PKG_DIR, "csrc", "knz_gpu.hip"), os.path.join(ROOT, "tests", "emu", "hip_emu.cpp"),
                               "-o", out])
    return out
