"""Schema-specialised kernels: source generation, hiprtc compile and the on-disk cache (no GPU needed:
hiprtc cross-compiles gfx950)."""
import os

import pytest

from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi


def test_generated_source_follows_the_schema_program():
    src = cabi.kernel_source(SCHEMAS["full"])
    assert '#include "spec_body.h"' in src and "rh_spec_size" in src and "rh_spec_emit" in src
    # full schema (scripts/generate_avro.py): 2 list loops (emails, phone_numbers), one 4-variant union, 2 nullable records
    assert src.count("h_list_begin<EMIT, CAREFUL>") == 2 and src.count("for (;;)") == 2
    assert src.count("h_union_begin<EMIT, CAREFUL>") == 1 and src.count("h_variant(") == 4
    assert src.count("h_rec_begin<EMIT, CAREFUL>") == 2 and src.count("h_rec_end(L)") == 2
    assert "static constexpr int K = 12, NDOM = 3" in src
    flat = cabi.kernel_source(SCHEMAS["flat4"])
    assert "static constexpr int K = 0, NDOM = 1" in flat and flat.count("h_fixed<EMIT, CAREFUL>") == 4


def test_prebuild_compiles_for_gfx950_and_caches(tmp_path, monkeypatch):
    monkeypatch.setenv("RUHVRO_HIP_KERNEL_CACHE", str(tmp_path))
    schema = SCHEMAS["t_enum"]
    assert cabi.prebuild(schema) is False            # compiled now
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(files) == 1
    blob = open(os.path.join(tmp_path, files[0]), "rb").read()
    assert blob[:4] == b"\x7fELF" and b"rh_spec_emit" in blob and b"gfx950" in blob
    assert cabi.prebuild(schema) is True             # cache hit
    assert cabi.prebuild(SCHEMAS["t_union"]) is False   # different schema -> different key
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2


def test_unsupported_schema_has_no_kernel():
    with pytest.raises(ValueError):
        cabi.kernel_source('{"type":"record","name":"B","fields":[{"name":"b","type":"bytes"}]}')
