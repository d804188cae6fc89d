"""The generic kernels' interpreter reads its program with SCALAR loads (kernels.hip `walk`, encode.hip `ECtx::walk`).

Through KParams' generic pointer the compiler fetched every op with two vector loads + v_readfirstlane behind s_waitcnt vmcnt(0)
-- a vector-L1 round trip per op and, in the emit walk, a drain of every store in flight (10M records of the benchmark schema:
4.21 ms; 3.39 ms with the constant-address-space fetch, profiles/r06_s5_generic_scalar_program.txt).  Nothing but the generated
code shows which of the two the compiler chose, so this test reads it (hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyruhvro_amd", "csrc")


def _functions(asm):
    """{kernel name: its instructions} of a --cuda-device-only -S listing."""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(rh_[a-z_]+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    return out


@pytest.mark.parametrize("src,kernels", [("kernels.hip", ("rh_k_size", "rh_k_emit")), ("encode.hip", ("rh_e_size", "rh_e_emit"))])
def test_interpreter_fetches_its_program_with_scalar_loads(tmp_path, src, kernels):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = os.path.join(tmp_path, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out], check=True, timeout=600)
    fns = _functions(open(out).read())
    for k in kernels:
        body = fns[k]
        x8 = [ln for ln in body if "s_load_dwordx8" in ln and ", 0x0" in ln]          # an op's first eight dwords
        x2 = [ln for ln in body if "s_load_dwordx2" in ln and ", 0x20" in ln]         # ... and its last two
        assert x8 and x2, (k, len(x8), len(x2))
        # the old fetch: a 12-byte vector load at offset 12 of an op (fields a, b, c) through a scalar base
        assert not [ln for ln in body if "global_load_dwordx3" in ln and "offset:12" in ln], k
