"""Build recipe for the native parts (in-tree, no pip install):

  libruhvro_hip.so   HIP kernels (gfx950) + engine + schema front-end, the C ABI of include/ruhvro_hip.h
  _pyruhvro.so       CPython extension: list[bytes] extraction under the GIL, Arrow C struct hand-off

hipcc cross-compiles gfx950 without a GPU, so this runs anywhere the ROCm toolchain is installed.
"""
from __future__ import annotations

import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libruhvro_hip.so")
EXT = os.path.join(HERE, "_pyruhvro.so")

LIB_SOURCES = ["kernels.hip", "engine.cpp", "schema.cpp"]
LIB_DEPS = LIB_SOURCES + ["program.h", "schema.h", "json.hpp", "../../include/ruhvro_hip.h", "../../include/arrow_c_abi.h"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if force or _stale(LIB, LIB_DEPS):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
               "-I", os.path.join(ROOT, "include")]
        cmd += [os.path.join(CSRC, s) for s in LIB_SOURCES]
        cmd += ["-o", LIB + ".tmp", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(LIB + ".tmp", LIB)
    return LIB


def build_ext(force: bool = False, verbose: bool = False) -> str:
    build_lib(force, verbose)
    if force or _stale(EXT, ["pymodule.cpp", "../../include/ruhvro_hip.h"]) or os.path.getmtime(EXT) < os.path.getmtime(LIB):
        inc = sysconfig.get_paths()["include"]
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", inc, "-I", os.path.join(ROOT, "include"),
               os.path.join(CSRC, "pymodule.cpp"), "-o", EXT + ".tmp",
               "-L", HERE, "-l:libruhvro_hip.so", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(EXT + ".tmp", EXT)
    return EXT


def build_all(force: bool = False, verbose: bool = False):
    return build_lib(force, verbose), build_ext(force, verbose)


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
