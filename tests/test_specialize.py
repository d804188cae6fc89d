"""Schema-specialised kernels: source generation, hiprtc compile and the on-disk cache (no GPU needed:
hiprtc cross-compiles gfx950)."""
import os
import subprocess
import sys

import pytest

from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi


def test_generated_source_follows_the_schema_program():
    src = cabi.kernel_source(SCHEMAS["full"])
    assert '#include "spec_body.h"' in src and "rh_spec_size" in src and "rh_spec_emit" in src
    # full schema (scripts/generate_avro.py): 2 list loops (emails, phone_numbers), one 4-variant union, 2 nullable records
    assert src.count("h_list_begin<EMIT, CAREFUL>") == 2 and src.count("for (;;)") == 2
    assert src.count("h_union_begin<EMIT, CAREFUL") == 1 and src.count("h_variant(") == 4
    assert src.count("h_rec_begin<EMIT, CAREFUL") == 2 and src.count("h_rec_end(L)") == 2
    assert "static constexpr int K = 12, KL = 12, NDOM = 3" in src
    # head fusion (walk.h read_head LA): the two record branch bytes open a 4-byte chain (2 | 4 << 2 = 18) that the record's
    # first string ends (1); the boolean in front of the union opens an 8-byte one (2 | 8 << 2 = 34) through the union index
    # (3) into the first head of each of its three non-null variants (1)
    # (round 6: a string's head is up to 3 bytes -- walk.h varint24 -- so the second record's branch byte + its nullable string
    #  need 5 bytes: an 8-byte chain, 34; the first one's branch byte + plain string stay within 4, 18)
    assert src.count("h_rec_begin<EMIT, CAREFUL, 18>") == 1 and src.count("h_rec_begin<EMIT, CAREFUL, 34>") == 1
    assert src.count("h_fixed<EMIT, CAREFUL, 34>") == 1
    assert src.count("h_union_begin<EMIT, CAREFUL, 3>") == 1 and src.count("CAREFUL, 1>(c, src, L, op)") == 5
    flat = cabi.kernel_source(SCHEMAS["flat4"])
    assert "static constexpr int K = 0, KL = 0, NDOM = 1" in flat and flat.count("h_fixed<EMIT, CAREFUL>") == 4


def test_prebuild_compiles_for_gfx950_and_caches(tmp_path, monkeypatch):
    monkeypatch.setenv("RUHVRO_HIP_KERNEL_CACHE", str(tmp_path))
    schema = SCHEMAS["t_enum"] + " "                 # (a schema handle of this test's own: handles remember their code objects)
    assert cabi.prebuild(schema) is False            # compiled now
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    # every kernel is its own code object (compiled side by side by rh_kcompile helper processes, kernel_jobs.cpp):
    # rh_spec_size / rh_spec_emit / rh_spec_fused, the ranged pair (tiles past the LDS window, round 6) and the Arrow -> Avro pair
    assert len(files) == 7
    assert sorted(os.listdir(tmp_path)) == sorted(files)      # no lock / source / log files left behind
    blobs = [open(os.path.join(tmp_path, f), "rb").read() for f in files]
    assert all(b[:4] == b"\x7fELF" and b"gfx950" in b for b in blobs)
    for entry in (b"rh_spec_fused", b"rh_espec_size", b"rh_espec_emit", b"rh_spec_size_r", b"rh_spec_emit_r"):
        assert sum(entry in b for b in blobs) == 1
    for entry in (b"rh_spec_size", b"rh_spec_emit"):                 # (also a prefix of the ranged pair's names)
        assert sum(entry in b for b in blobs) == 2
    assert cabi.prebuild(schema) is True             # nothing to compile
    assert cabi.prebuild(SCHEMAS["t_enum"] + "  ") is True     # another handle of the same schema: disk cache hit
    assert cabi.prebuild(SCHEMAS["t_union"] + " ") is False   # different schema -> different keys
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 14
    assert cabi.kernels_ready(schema) and cabi.kernels_ready(schema, encode=True)


def test_in_process_compile_when_the_helper_is_absent(tmp_path):
    """Without rh_kcompile next to the library the jobs compile on threads of the process (RUHVRO_HIP_KCOMPILE=0 forces it)."""
    code = ("import os, sys; from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS\n"
            "assert cabi.prebuild(SCHEMAS['t_enum']) is False\n"
            "assert len([f for f in os.listdir(os.environ['RUHVRO_HIP_KERNEL_CACHE']) if f.endswith('.hsaco')]) == 7\n")
    env = dict(os.environ, RUHVRO_HIP_KERNEL_CACHE=str(tmp_path), RUHVRO_HIP_KCOMPILE="0")
    subprocess.check_call([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_compile_failure_is_reported_not_hung(tmp_path):
    """A helper that cannot compile (here: one that always fails) turns into an error message, for prebuild and kernels_ready."""
    fake = tmp_path / "fake_kcompile"
    fake.write_text("#!/bin/sh\necho 'no compiler today' > \"$3\"\nexit 1\n")
    fake.chmod(0o755)
    code = ("from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS\n"
            "try:\n    cabi.prebuild(SCHEMAS['t_enum']); raise SystemExit('no error')\n"
            "except RuntimeError as e:\n    assert 'no compiler today' in str(e), str(e)\n")
    env = dict(os.environ, RUHVRO_HIP_KERNEL_CACHE=str(tmp_path / "kc"), RUHVRO_HIP_KCOMPILE=str(fake))
    subprocess.check_call([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_encode_source_issues_a_row_domains_loads_before_its_first_write():
    """The specialised Arrow -> Avro walk is the schema program unrolled over encode_walk.h's handlers with the
    e_*_load half of every field of a row domain hoisted in front of the first e_*_put (latency-bound walk)."""
    src = cabi.encode_kernel_source(SCHEMAS["full"])
    body = src[src.index("static __device__ __forceinline__ void walk"):]
    first_put = body.index("_put<MODE>")
    dom0_loads = [ln for ln in body[:first_put].splitlines() if "_load(c, op" in ln]
    assert len(dom0_loads) >= 15                     # every dom-0 field of the generate_avro.py schema
    assert "rh_espec_size" in src and "rh_espec_emit" in src
    loop = body[body.index("for (;;) {"):]
    assert loop.index("e_list_next") < loop.index("e_span_load") < loop.index("e_string_put")   # list items: loads per iteration
    # enum symbols are folded into the kernel as (length, masked dword) compares
    assert "n == 1u && (a.x & 0x000000ffu) == 0x00000041u" in src
    # SURVEY 8f N4 in this direction: bytes is the string leaf, fixed / decimal / uuid have their own handler pair
    n4 = cabi.encode_kernel_source('{"type":"record","name":"B","fields":[{"name":"b","type":"bytes"},'
                                   '{"name":"f","type":{"type":"fixed","name":"f7","size":7}}]}')
    assert "e_string_put_cf<MODE>" in n4 and "e_bin_load(c, op" in n4 and "e_bin_put<MODE>" in n4
    with pytest.raises(ValueError):
        cabi.encode_kernel_source('{"type":"record","name":"B","fields":[{"name":"b","type":{"type":"long","logicalType":"timestamp-nanos"}}]}')


def test_encode_list_bodies_are_software_pipelined_with_a_guarded_look_ahead(monkeypatch):
    """Round 4: inside a list loop the first-level loads of the NEXT item are issued at the top of an iteration, and a top-level
    list's first item is requested in front of the loop with the row's string fetches.  The look-ahead is per lane and only
    taken with two or more items left: row + 1 may lie behind the column, where an offsets slot holds padding (that read
    faulted once the arena pool handed out used memory: profiles/r04zg_*)."""
    src = cabi.encode_kernel_source(SCHEMAS["full"])
    body = src[src.index("static __device__ __forceinline__ void walk"):]
    # emails (one string per item) and phone_numbers (key + value): three look-ahead loads, each guarded, each preloaded once
    assert body.count("c.rem(0) > 1u ? 1u : 0u") == 3
    assert body.count("= e_span_load(c, op4, v2.s0);") == 1 and body.count("= e_span_load(c, op14, v12.s0);") == 1
    loop = body[body.index("for (;;) {"):]
    assert loop.index("const SpanV v4 = nv4;") < loop.index("nv4 = e_span_load(c, op4,") < loop.index("e_string_fetch<MODE>(c, op4, v4)")
    assert ", 1u)" not in body                          # never an unconditional row + 1
    monkeypatch.setenv("RUHVRO_HIP_NO_ENC_PIPE", "1")    # the A/B knob generates the loop of round 3
    old = cabi.encode_kernel_source(SCHEMAS["full"])
    assert "nv4" not in old and old.count("for (;;) {") == 2


def test_unsupported_schema_has_no_kernel():
    with pytest.raises(ValueError):
        cabi.kernel_source('{"type":"record","name":"B","fields":[{"name":"b","type":{"type":"long","logicalType":"local-timestamp-micros"}}]}')
    assert "h_bin<EMIT, CAREFUL>" in cabi.kernel_source('{"type":"record","name":"B","fields":[{"name":"b","type":{"type":"fixed","name":"f","size":7}}]}')


STAGED_VARIANTS = "SOME_EXPERIMENT,OTHER_1"


def test_staged_kernel_variants_compile_and_are_keyed_apart(tmp_path, monkeypatch):
    """RUHVRO_HIP_VARIANT only prepends `#define RH_V_<NAME> 1` lines to the generated source (experimental code
    paths staged in the device headers for an A/B on the GPU).  They must at least compile for
    gfx950, live under their own cache keys, and leave the default source untouched."""
    monkeypatch.setenv("RUHVRO_HIP_KERNEL_CACHE", str(tmp_path))
    schema = SCHEMAS["array_and_map"]
    monkeypatch.delenv("RUHVRO_HIP_VARIANT", raising=False)
    base_src = cabi.kernel_source(schema)
    assert "RH_V_" not in base_src
    assert cabi.prebuild(schema) is False
    n_default = len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")])
    monkeypatch.setenv("RUHVRO_HIP_VARIANT", STAGED_VARIANTS)
    var_src = cabi.kernel_source(schema)
    for name in STAGED_VARIANTS.split(","):
        assert f"#define RH_V_{name} 1" in var_src
    assert var_src.replace("".join(f"#define RH_V_{n} 1\n" for n in STAGED_VARIANTS.split(",")), "") == base_src
    # (a schema handle remembers its code objects; the variant set is read per generated source -> a fresh handle sees it)
    assert cabi.prebuild(schema + " ") is False                # compiled now, under other keys
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2 * n_default
    monkeypatch.setenv("RUHVRO_HIP_VARIANT", "not a name,lower,OK_1")
    assert "RH_V_OK_1" in cabi.kernel_source(schema) and "lower" not in cabi.kernel_source(schema)


def _kernel_notes(hsaco_path):
    """{kernel name: {metadata key: int}} from the code object's AMDGPU notes."""
    import re
    import subprocess
    exe = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(exe):
        pytest.skip("llvm-readelf not available")
    txt = subprocess.run([exe, "--notes", hsaco_path], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s*\.(\w+):\s+(\d+)\s*$", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def test_specialised_kernels_of_the_benchmark_schema_stay_in_registers(tmp_path, monkeypatch):
    """Resource guard for the headline kernels (BASELINE config 4 schema): no scratch, no VGPR spills, VGPR counts
    that keep 4 workgroups of 4 waves per CU resident (LDS-bound), and the emit kernel's SGPR spills well under the
    194 it had while the 40 buffer addresses lived in SGPRs (every spill reload is a VALU instruction and the kernel
    is VALU-bound -- DESIGN.md section 5)."""
    # Compiled in a FRESH interpreter, like build() does for the kernels that ship: a process that has imported torch
    # resolves libhiprtc to the older one bundled with torch (ROCm 7.0), whose register allocation differs (it leaves
    # 16 bytes of scratch in rh_spec_emit) -- the guard is for the code objects of the image's own hiprtc.
    env = {k: v for k, v in os.environ.items() if k not in ("RUHVRO_HIP_VARIANT", "RUHVRO_HIP_PROFILE", "RUHVRO_HIP_TILE")}
    env["RUHVRO_HIP_KERNEL_CACHE"] = str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; "
                    "assert cabi.prebuild(SCHEMAS['full']) is not None"], check=True, cwd=root, env=env, timeout=600)
    notes = {}
    for f in os.listdir(tmp_path):
        if f.endswith(".hsaco"):
            notes.update(_kernel_notes(os.path.join(tmp_path, f)))
    assert set(notes) >= {"rh_spec_size", "rh_spec_emit", "rh_espec_size", "rh_espec_emit"}
    for name, md in notes.items():
        assert md["vgpr_count"] <= 128, (name, md)
        if name.endswith("_r"):          # the ranged pair (tiles past the LDS window): a cold path, its item scan keeps four context copies
            continue
        assert md["private_segment_fixed_size"] == 0, (name, md)
        assert md["vgpr_spill_count"] == 0, (name, md)
    assert notes["rh_spec_emit"]["sgpr_spill_count"] <= 110, notes["rh_spec_emit"]
    assert notes["rh_spec_size"]["sgpr_spill_count"] == 0 and notes["rh_espec_emit"]["sgpr_spill_count"] == 0


def test_processes_that_meet_a_new_schema_together_compile_it_once(tmp_path):
    """Eight ranks of one job meet a new schema at the same moment (BASELINE config 5): every process starts its compile helpers,
    which serialise on a lock file next to each code object -- one compiles, the others find the object there -- and every process
    ends up with the seven kernels (size / emit / single-pass, the ranged pair, the Arrow -> Avro pair), nothing left behind."""
    code = ("import os; from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS\n"
            "cabi.prebuild(SCHEMAS['t_union'])\n"
            "assert cabi.kernels_ready(SCHEMAS['t_union']) and cabi.kernels_ready(SCHEMAS['t_union'], encode=True)\n")
    env = dict(os.environ, RUHVRO_HIP_KERNEL_CACHE=str(tmp_path), RUHVRO_HIP_COMPILE_JOBS="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, cwd=root) for _ in range(4)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0, 0, 0]
    left = sorted(os.listdir(tmp_path))
    assert len(left) == 7 and all(f.endswith(".hsaco") for f in left), left


def test_unwritable_kernel_cache_falls_back_to_a_private_directory(tmp_path):
    """An installation whose kernel cache cannot be written (a read-only site-packages; here: a path that is a FILE, which stops
    root too) still compiles: the jobs work in a private temporary directory of the process, and the caller gets its kernels."""
    blocked = tmp_path / "not_a_directory"
    blocked.write_text("x")
    code = ("from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS\n"
            "assert cabi.prebuild(SCHEMAS['t_enum']) is False\n"
            "assert cabi.kernels_ready(SCHEMAS['t_enum'])\n")
    env = dict(os.environ, RUHVRO_HIP_KERNEL_CACHE=str(blocked))
    subprocess.check_call([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert blocked.read_text() == "x"
