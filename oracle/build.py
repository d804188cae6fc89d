"""ORACLE build recipe: compiles oracle_walk.c (our C restatement) with gcc.

The reference itself is Rust (cargo/rustc absent from this image), so there is
no ``oracle/_ref`` build: the reference cannot be compiled here -- see DESIGN.md.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle_walk.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "oracle_walk.c")
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    cmd = ["gcc", "-O3", "-march=x86-64-v2", "-shared", "-fPIC", "-pthread", "-Wall", src, "-o", LIB + ".tmp"]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
