"""device-resident output through the Arrow C *Device* Data interface (rh_device_result_export).
Every chunk is exported as an ArrowDeviceArray, its ArrowArray tree is walked with ctypes next to the host copy of
the same chunk, every buffer is copied back with hipMemcpy and compared byte for byte (bitmaps bit-masked) with
to_host() and with the oracle; views are released and re-exported to prove the ownership rules of the header
("valid until rh_device_result_free; release the view through array.release")."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

from arrow_compare import assert_batches_identical
from avrogen import synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker

from pyruhvro_amd import cabi
import hipmem

pytestmark = pytest.mark.gpu

ARROW_DEVICE_ROCM = 10
RELEASE_FN = C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))


def _d2h(hip, dptr, n):
    return hipmem.d2h(dptr, n)


def _mask_bits(a, nbits):
    a = a.copy()
    if nbits & 7 and len(a):
        a[-1] &= (1 << (nbits & 7)) - 1
    return a


def _host(buf, n):
    return np.frombuffer(buf, dtype=np.uint8, count=n) if n else np.zeros(0, dtype=np.uint8)


def _walk(hip, dev, host: pa.Array, path, seen):
    """dev: ctypes ArrowArray whose buffers are DEVICE pointers; host: the same node from to_host()."""
    t = host.type
    n = len(host)
    assert dev.length == n and dev.offset == 0 and dev.null_count == host.null_count, path
    hb = host.buffers()
    bufs = [dev.buffers[i] for i in range(dev.n_buffers)]
    for p in bufs:
        if p:
            seen.append(p)
    if pa.types.is_null(t):
        assert dev.n_buffers == 0 and dev.n_children == 0
        return
    if not pa.types.is_union(t):
        assert (bufs[0] is None or bufs[0] == 0) == (hb[0] is None), f"{path}: validity presence"
        if hb[0] is not None:
            nb = (n + 7) // 8
            assert np.array_equal(_mask_bits(_d2h(hip, bufs[0], nb), n), _mask_bits(_host(hb[0], nb), n)), f"{path}: validity"
    if pa.types.is_boolean(t):
        nb = (n + 7) // 8
        assert np.array_equal(_mask_bits(_d2h(hip, bufs[1], nb), n), _mask_bits(_host(hb[1], nb), n)), f"{path}: bool values"
    elif pa.types.is_string(t):
        assert dev.n_buffers == 3
        off = _d2h(hip, bufs[1], 4 * (n + 1))
        assert np.array_equal(off, _host(hb[1], 4 * (n + 1))), f"{path}: offsets"
        last = int(off.view(np.int32)[-1])
        assert np.array_equal(_d2h(hip, bufs[2], last), _host(hb[2], last)), f"{path}: string bytes"
    elif pa.types.is_struct(t):
        assert dev.n_children == t.num_fields
        for i in range(t.num_fields):
            _walk(hip, dev.children[i].contents, host.field(i), f"{path}.{t.field(i).name}", seen)
    elif pa.types.is_union(t):
        assert dev.n_buffers == 1 and dev.n_children == t.num_fields          # sparse: type ids only
        assert np.array_equal(_d2h(hip, bufs[0], n), _host(hb[1], n)), f"{path}: type ids"
        for i in range(t.num_fields):
            _walk(hip, dev.children[i].contents, host.field(i), f"{path}<{i}>", seen)
    elif pa.types.is_map(t):
        assert np.array_equal(_d2h(hip, bufs[1], 4 * (n + 1)), _host(hb[1], 4 * (n + 1))), f"{path}: map offsets"
        entries = dev.children[0].contents
        assert entries.n_children == 2 and entries.length == len(host.keys) and entries.null_count == 0
        _walk(hip, entries.children[0].contents, host.keys, f"{path}.keys", seen)
        _walk(hip, entries.children[1].contents, host.items, f"{path}.values", seen)
    elif pa.types.is_list(t):
        assert np.array_equal(_d2h(hip, bufs[1], 4 * (n + 1)), _host(hb[1], 4 * (n + 1))), f"{path}: list offsets"
        _walk(hip, dev.children[0].contents, host.values, f"{path}[]", seen)
    else:
        w = t.bit_width // 8
        assert np.array_equal(_d2h(hip, bufs[1], w * n), _host(hb[1], w * n)), f"{path}: values"


def _decode_device(recs, schema, k, kernel):
    data, offsets = c_walker.pack(recs)
    d_data, d_off = hipmem.upload_packed(data, offsets)
    r = cabi.decode_device(d_data.ptr, d_off.ptr, int(offsets[-1]), len(recs), schema, k, device=0, kernel=kernel)
    return r, (d_data, d_off)


@pytest.mark.parametrize("kernel", [cabi.KERNEL_GENERIC, cabi.KERNEL_SPECIALIZED], ids=["generic", "specialized"])
@pytest.mark.parametrize("name,n,k", [("full", 3000, 4), ("cfg3", 1000, 3), ("array_and_map", 500, 2), ("flat4", 100, 1), ("full", 0, 1)])
def test_arrow_device_array_export(name, n, k, kernel):
    hip = hipmem.hip()
    L = cabi.lib()
    recs = synth.records(name, n, seed=3) if n else []
    r, keep = _decode_device(recs, SCHEMAS[name], k, kernel)
    host = r.to_host()
    exp = c_walker.decode_threaded(recs, SCHEMAS[name], k)
    assert len(host) == len(exp) == r.chunks
    for h, e in zip(host, exp):
        assert_batches_identical(h, e)
    views = []
    for rnd in range(2):                                   # export, check, release -- then once more: ownership is sound
        for c in range(r.chunks):
            d = cabi.ArrowDeviceArray()
            assert L.rh_device_result_export(r.handle, c, C.byref(d)) == 0
            assert d.device_type == ARROW_DEVICE_ROCM and d.device_id == 0 and not d.sync_event
            assert d.array.n_buffers == 1 and not d.array.buffers[0]          # the batch as a struct array: no validity
            sa = host[c].to_struct_array()
            seen = []
            assert d.array.length == len(sa) and d.array.n_children == sa.type.num_fields
            for i in range(sa.type.num_fields):
                _walk(hip, d.array.children[i].contents, sa.field(i), sa.type.field(i).name, seen)
            if n and name == "full":                       # the buffers really are device memory of device 0
                attr = (C.c_byte * 256)()
                assert hip.hipPointerGetAttributes(attr, seen[0]) == 0
            if rnd == 0 and c == 0:
                views.append(d)                            # one view stays alive across the other exports
            else:
                RELEASE_FN(d.array.release)(C.byref(d.array))
                assert not d.array.release                 # released views are marked (Arrow C ABI)
    assert L.rh_device_result_export(r.handle, r.chunks, C.byref(cabi.ArrowDeviceArray())) != 0    # chunk out of range
    for d in views:
        RELEASE_FN(d.array.release)(C.byref(d.array))
    again = r.to_host()                                    # the result is intact after the views are gone
    for h, e in zip(again, exp):
        assert_batches_identical(h, e)
    r.free()
