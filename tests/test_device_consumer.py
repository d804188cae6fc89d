"""SURVEY.md 8f N3, the consumer side: pyruhvro_amd.deserialize_to_device leaves the Arrow buffers in HBM and hands every one
of them out through the DLPack protocol.  torch is the consumer here (torch.from_dlpack): it computes on the engine's buffers in
place, and what it computes equals what the oracle's host batches say.  Needs an MI355X."""
import gc
import weakref

import numpy as np
import pyarrow as pa
import pytest
import torch

from arrow_compare import assert_batches_identical
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from oracle import c_walker

import pyruhvro_amd as P
from pyruhvro_amd import device as D

pytestmark = pytest.mark.gpu

N, K = 200_003, 8


@pytest.fixture(scope="module")
def decoded():
    data, offsets = fastgen.generate("full", N)
    recs = fastgen.split(data, offsets)
    exp = c_walker.decode_threaded(recs, SCHEMAS["full"], K)
    return recs, exp


def test_torch_computes_on_the_engines_buffers_in_place(decoded):
    recs, exp = decoded
    dec = P.deserialize_to_device(recs, SCHEMAS["full"], K)
    assert len(dec.batches) == K and sum(b.num_rows for b in dec.batches) == N
    total = 0
    for b, e in zip(dec.batches, exp):
        col = b.column("created_at")
        assert col.type == pa.int64() and col.null_count == 0 and col.validity is None
        t = torch.from_dlpack(col.values)
        assert t.dtype == torch.int64 and t.device.type == "cuda" and t.numel() == e.num_rows
        assert t.data_ptr() == col.values.ptr                                   # the engine's memory itself: no copy
        total += int(t.sum())
        assert torch.equal(t.cpu(), torch.from_numpy(e.column("created_at").to_numpy().copy()))
    assert total == sum(int(np.sum(e.column("created_at").to_numpy())) for e in exp)
    # the whole result, copied out, is still the oracle's
    for g, e in zip(dec.to_host(), exp):
        assert_batches_identical(g, e)


def test_every_buffer_role_is_a_typed_view(decoded):
    recs, exp = decoded
    dec = P.deserialize_to_device(recs, SCHEMAS["full"], K)
    b, e = dec.batches[3], exp[3]
    # a nullable string column: validity bitmap + int32 offsets + data bytes
    name, en = b.column("name"), e.column("name")
    off = torch.from_dlpack(name.offsets)
    assert off.dtype == torch.int32 and off.numel() == e.num_rows + 1 and int(off[0]) == 0
    assert int(off[-1]) == len(en.buffers()[2]) if False else int(off[-1]) == int(np.frombuffer(en.buffers()[1], np.int32, e.num_rows + 1)[-1])
    dat = torch.from_dlpack(name.data)
    assert dat.dtype == torch.uint8 and dat.numel() >= int(off[-1])
    assert bytes(dat[: int(off[-1])].cpu().numpy()) == en.buffers()[2].to_pybytes()[: int(off[-1])]
    assert name.null_count == en.null_count > 0
    val = torch.from_dlpack(name.validity).cpu().numpy()
    assert bytes(val[: (e.num_rows + 7) // 8]) == en.buffers()[0].to_pybytes()[: (e.num_rows + 7) // 8]
    # nullable int32
    age = torch.from_dlpack(b.column("age").values)
    assert age.dtype == torch.int32 and torch.equal(age.cpu(), torch.from_numpy(np.frombuffer(e.column("age").buffers()[1], np.int32, e.num_rows).copy()))
    # list<string>: offsets on the list, string child in the item domain
    emails = b.column("emails")
    lo = torch.from_dlpack(emails.offsets)
    item = emails.children[0]
    assert item.length == int(lo[-1]) == len(e.column("emails").values)
    assert int(torch.from_dlpack(item.offsets)[-1]) == sum(len(s) for s in e.column("emails").values.to_pylist())
    # map<string, string>: entries struct with keys / values
    ent = b.column("phone_numbers").children[0]
    mt = b.schema.field("phone_numbers").type
    assert [c.name for c in ent.children] == ["keys", mt.item_field.name] and ent.length == len(e.column("phone_numbers").keys)
    assert int(torch.from_dlpack(ent.children[1].offsets)[-1]) == sum(len(s) for s in e.column("phone_numbers").items.to_pylist())
    # struct with a boolean child (a bitmap) and a sparse union with its type ids
    news = b.column("preferences").child("newsletter")
    assert news.values.dtype == np.uint8 and news.values.nbytes >= (e.num_rows + 7) // 8
    tid = torch.from_dlpack(b.column("status").buffers["type_ids"])
    assert tid.dtype == torch.int8 and torch.equal(tid.cpu(), torch.from_numpy(e.column("status").type_codes.to_numpy().copy()))
    assert len(b.column("status").children) == 4


def test_views_keep_the_result_alive_and_release_it(decoded):
    recs, exp = decoded
    dec = P.deserialize_to_device(recs[:50_000], SCHEMAS["full"], 2)
    owner = weakref.ref(dec._result)
    want = int(np.sum(c_walker.decode_threaded(recs[:50_000], SCHEMAS["full"], 2)[0].column("created_at").to_numpy()))
    t = torch.from_dlpack(dec.batches[0].column("created_at").values)
    cap = dec.batches[0].column("age").values.__dlpack__()       # a capsule nobody consumes
    del dec
    gc.collect()
    assert owner() is not None                                    # the tensor (and the unconsumed capsule) hold the device memory
    assert int(t.sum()) == want
    del cap
    gc.collect()
    assert owner() is not None and int(t.sum()) == want
    del t
    gc.collect()
    torch.cuda.synchronize()
    assert owner() is None and not D._live                        # last view gone: the result (and its arena) is released


def test_device_resident_input_and_errors(decoded):
    recs, exp = decoded
    data, offsets = fastgen.generate("full", 40_000)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda")
    dec = P.deserialize_to_device((d_data, d_off), SCHEMAS["full"], 4, stream=torch.cuda.current_stream().cuda_stream)
    e4 = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, 4, threaded=True)
    for g, e in zip(dec.to_host(), e4):
        assert_batches_identical(g, e)
    arr = pa.array(recs[:1000], type=pa.binary())
    dec = P.deserialize_to_device(arr, SCHEMAS["full"], 3)
    for g, e in zip(dec.to_host(), c_walker.decode_threaded(recs[:1000], SCHEMAS["full"], 3)):
        assert_batches_identical(g, e)
    bad = list(recs[:100])
    bad[57] = bad[57][:5]
    with pytest.raises(ValueError, match="unexpected end of buffer"):
        P.deserialize_to_device(bad, SCHEMAS["full"], 2)
    with pytest.raises(ValueError):
        P.deserialize_to_device(recs[:10], '{"type": "record", "name": "x", "fields": [{"name": "a", "type": "bytes_"}]}', 1)


def test_duration_columns_are_int64_views():
    """An Avro duration (Arrow Duration(ms), schema.cpp format `tDm`) is an int64 buffer like a timestamp: the device view hands it
    out as such, top level, nullable, and as a list item (`_value_dtype` had no case for it)."""
    import test_n4_types as N4
    from oracle import py_walker
    recs = N4._dur_records(300)
    exp = py_walker.decode(recs, N4.DUR_SCHEMA, extended=True)
    dec = P.deserialize_to_device(recs, N4.DUR_SCHEMA, 1)
    b = dec.batches[0]
    for name in ("d", "nd"):
        col = b.column(name)
        assert col.type == pa.duration("ms")
        t = torch.from_dlpack(col.values)
        assert t.dtype == torch.int64 and t.numel() == 300
        assert t.cpu().tolist() == exp.column(name).cast(pa.int64()).fill_null(0).to_pylist()
    item = b.column("arr").children[0]
    assert item.type == pa.duration("ms")
    assert torch.from_dlpack(item.values).cpu().tolist() == exp.column("arr").values.cast(pa.int64()).to_pylist()
    for g in dec.to_host():
        assert_batches_identical(g, exp)
