"""The engine under AddressSanitizer + UndefinedBehaviorSanitizer (`_build.build_sanitized`).

The engine (csrc/engine_*.cpp) manages lifetimes by hand (memory pools, leases, turnstiles, Arrow release callbacks shared by k chunks,
malloc'd error strings).  A child interpreter preloads the ASan runtime and loads the sanitizer build of
libruhvro_hip.so in place of the normal one (RUHVRO_HIP_LIB for the ctypes view, LD_LIBRARY_PATH for the CPython
extension's NEEDED entry), then runs
  * on any box: the host logic -- schema compile / export / free for every known schema, generated kernel source,
    bad schemas, the shard deal, rh_encode's batch binder and every no-device failure path;
  * on the GPU box (-m gpu): the export / ownership tests and the multi-shard driver, i.e. the whole path.
Any sanitizer report aborts the child (halt_on_error) and fails the test."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

HOST_WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    ROOT = %r
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pyarrow as pa
    import pyruhvro_amd as P
    from pyruhvro_amd import cabi
    assert "_san" in cabi.LIB_PATH and "_san/libruhvro_hip.so" in open("/proc/self/maps").read()
    from known_schemas import known_schemas
    L = cabi.lib()
    n = 0
    for sj in dict.fromkeys(known_schemas()):
        raw = sj.encode(); err = C.c_char_p()
        h = L.rh_schema_compile(raw, len(raw), C.byref(err))
        if not h:
            cabi._take_err(err); continue
        cs = cabi.ArrowSchema()
        assert L.rh_schema_export(h, C.byref(cs)) == 0
        pa.DataType._import_from_c(C.addressof(cs))          # pyarrow calls our release callback
        p = L.rh_schema_kernel_source(h); assert p; L.rh_free_string(p)
        p = L.rh_schema_encode_kernel_source(h); assert p; L.rh_free_string(p)
        L.rh_schema_free(h); n += 1
    assert n > 50
    for bad in ("{", '{"type":"record","name":"B","fields":[{"name":"b","type":"bytes_typo"}]}', '"string"', ""):
        try:
            cabi.Schema(bad); raise SystemExit("accepted " + bad)
        except ValueError:
            pass
    assert [cabi.shard_chunks(103, 8, 3, j) for j in range(3)] == [(0, 2, 0, 24), (2, 5, 24, 60), (5, 8, 60, 103)]
    from avrogen.schemas import SCHEMAS
    rb = pa.RecordBatch.from_arrays([pa.array([1], pa.int32()), pa.array([2], pa.int64()), pa.array([0.5]), pa.array([True])], names=["i", "l", "d", "b"])
    for batch in (rb.drop_columns(["d"]), rb.set_column(0, "i", pa.array([1], pa.int64()))):
        try:
            P.serialize_record_batch(batch, SCHEMAS["flat4"], 1); raise SystemExit("bound a mismatched batch")
        except ValueError:
            pass
    if P.device_count() == 0:            # every decode / encode entry point fails loudly, and cleanly, without a device
        import numpy as np
        for call in (lambda: P.deserialize_array_threaded([b"\\x00"] * 9, SCHEMAS["flat4"], 3),
                     lambda: cabi.decode_packed(np.zeros(4, np.uint8), np.array([0, 2, 4], np.uint64), SCHEMAS["flat4"], 2, devices=[0, 0]),
                     lambda: cabi.decode_device(0, 0, 0, 0, SCHEMAS["flat4"], 1),
                     lambda: P.serialize_record_batch(rb, SCHEMAS["flat4"], 1)):
            try:
                call(); raise SystemExit("decoded without a device")
            except RuntimeError:
                pass
    print("SANITIZED_HOST_OK", n)
""")


def _libstdcxx():
    out = subprocess.check_output(["gcc", "-print-file-name=libstdc++.so.6"], text=True).strip()
    return os.path.realpath(out) if os.path.isabs(out) else "libstdc++.so.6"


def _san_env():
    from pyruhvro_amd._build import SAN_DIR, asan_runtime, build_sanitized
    rt = asan_runtime()
    if not rt:
        pytest.skip("no shared gcc ASan runtime on this box")
    lib = build_sanitized()
    env = dict(os.environ)
    env.update({
        # libstdc++ next to it: python does not link it, and ASan's __cxa_throw interceptor needs the real one at start-up
        "LD_PRELOAD": rt + " " + _libstdcxx(), "RUHVRO_HIP_LIB": lib,
        "LD_LIBRARY_PATH": SAN_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", ""),
        # python itself leaks by design; the GPU driver maps memory ASan's shadow gap check trips over
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "RUHVRO_HIP_KERNEL_CACHE": os.path.join(ROOT, "pyruhvro_amd", "_kcache"),
    })
    return env


def test_host_logic_under_asan_ubsan(tmp_path):
    env = _san_env()
    script = tmp_path / "w.py"
    script.write_text(HOST_WORKER % ROOT)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_HOST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]


@pytest.mark.gpu
def test_gpu_path_under_asan_ubsan():
    """Device export / release / re-export, the multi-shard driver with its error path, and a slice decode, with the
    sanitized engine.  (The HIP runtime itself is not instrumented; only reports that stop the child count.)"""
    env = _san_env()
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_device_export.py", "tests/test_multi_gpu.py",
                        "tests/test_gpu_parity.py::test_chunk_semantics", "tests/test_gpu_parity.py::test_input_forms",
                        "tests/test_gpu_encode.py::test_chunking_and_empty_batches"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
