// Arrow -> Avro on the GPU (rh_encode / rh_encode_device): binds the batch's buffers to the schema program, runs the size /
// scan / emit kernels of encode.hip (or their specialised forms), exports BinaryArrays.
#include "engine_internal.h"

using namespace rhe;

// ===========================================================================
// Arrow -> Avro: rh_encode
// ===========================================================================

namespace {

using EncodeError = ValueClassError;

struct InBuf {            // logical range of one input buffer (host side), rebased to row 0
  const uint8_t* host = nullptr;
  uint64_t bytes = 0;
  uint32_t bitoff = 0;
};

struct StrSrc {           // where a string / enum node's text lives on the host (for error messages)
  const int32_t* offsets = nullptr;
  const uint8_t* data = nullptr;
};

// Mirrors build_record_encoder / build_field_encoder / build_union_encoder / build_nullable_encoder
// (ruhvro/src/fast_encode.rs:151-358): walks the Avro type tree, the decoder nodes built from it and the
// Arrow C Data structs in lockstep, matching record fields to struct children BY NAME.
struct EncodeBinder {
  const CompiledSchema& cs;
  bool device_ptrs = false;   // rh_encode_device: the batch's buffer pointers are device pointers (never dereferenced here)
  std::vector<InBuf> in;
  std::vector<StrSrc> strs;   // by node id
  uint64_t max_rows = 0;      // longest array bound (sizes the shared all-ones validity bitmap)

  explicit EncodeBinder(const CompiledSchema& c) : cs(c), in(c.bufs.size()), strs(c.nodes.size()) {}

  static const rh::AvroType* null_inner(const rh::AvroType& u) {
    if (u.variants.size() != 2) return nullptr;
    if (u.variants[0]->kind == rh::AV_NULL) return u.variants[1].get();
    if (u.variants[1]->kind == rh::AV_NULL) return u.variants[0].get();
    return nullptr;
  }

  void validity(int buf, const ArrowArray* a, int64_t off, int64_t len) {
    if (buf < 0) return;
    if (a->n_buffers < 1 || !a->buffers[0] || a->null_count == 0) return;   // absent = all valid
    in[buf].host = (const uint8_t*)a->buffers[0] + (off >> 3);
    in[buf].bitoff = (uint32_t)(off & 7);
    in[buf].bytes = (uint64_t)((in[buf].bitoff + len + 7) >> 3);
  }

  void bind(const rh::AvroType& t0, int id, const ArrowSchema* fs, const ArrowArray* fa, int64_t off, int64_t len) {
    const rh::AvroType* t = &t0;
    if (t->kind == rh::AV_UNION)
      if (const rh::AvroType* inner = null_inner(*t)) t = inner;     // 2-variant null union: the node is the inner type, nullable
    const DecNode& n = cs.nodes[id];
    const std::string fmt = fs->format ? fs->format : "";
    if (len < 0 || off < 0) throw EncodeError("fast_encode: arrow array downcast failed");
    max_rows = std::max<uint64_t>(max_rows, (uint64_t)len);
    switch (n.kind) {
      case rh::NK_NULL:
        return;
      case rh::NK_FIXED: {
        static const char* want[] = {"i", "l", "f", "g", "b"};
        bool ok = fmt == want[n.fixed];
        if (t->kind == rh::AV_DATE) ok = fmt == "tdD";
        if (t->kind == rh::AV_TS_MILLIS) ok = fmt.rfind("tsm:", 0) == 0;
        if (t->kind == rh::AV_TS_MICROS) ok = fmt.rfind("tsu:", 0) == 0;
        if (t->kind == rh::AV_TIME_MILLIS) ok = fmt == "ttm";
        if (t->kind == rh::AV_TIME_MICROS) ok = fmt == "ttu";
        if (!ok || fa->n_buffers < 2 || (len > 0 && !fa->buffers[1])) throw EncodeError("fast_encode: arrow array downcast failed");
        InBuf& v = in[n.buf_main];
        if (n.fixed == rh::FK_BOOL) {
          v.host = (const uint8_t*)fa->buffers[1] + (off >> 3);
          v.bitoff = (uint32_t)(off & 7);
          v.bytes = (uint64_t)((v.bitoff + len + 7) >> 3);
        } else {
          const uint64_t w = (n.fixed == rh::FK_I32 || n.fixed == rh::FK_F32) ? 4 : 8;
          v.host = (const uint8_t*)fa->buffers[1] + (uint64_t)off * w;
          v.bytes = (uint64_t)len * w;
        }
        if (len == 0) v.host = nullptr;
        if (n.nullable) validity(n.buf_validity, fa, off, len);
        return;
      }
      case rh::NK_BIN: {              // FixedSizeBinary(N) / Decimal128 values, `bin_width` bytes per row
        const std::string want = t->kind == rh::AV_DECIMAL    ? "d:" + std::to_string(t->precision) + "," + std::to_string(t->scale)
                                 : t->kind == rh::AV_DURATION ? "tDm"
                                                              : "w:" + std::to_string(n.bin_width);
        if (fmt != want || fa->n_buffers < 2 || (len > 0 && !fa->buffers[1])) throw EncodeError("fast_encode: arrow array downcast failed");
        InBuf& v = in[n.buf_main];
        v.host = len ? (const uint8_t*)fa->buffers[1] + (uint64_t)off * (uint64_t)n.bin_width : nullptr;
        v.bytes = (uint64_t)len * (uint64_t)n.bin_width;
        if (n.nullable) validity(n.buf_validity, fa, off, len);
        return;
      }
      case rh::NK_STRING:
      case rh::NK_ENUM: {
        if (fmt != (t->kind == rh::AV_BYTES ? "z" : "u") || fa->n_buffers < 3) throw EncodeError("fast_encode: arrow array downcast failed");
        if (fa->buffers[1]) {           // an empty array may come without an offsets buffer
          const int32_t* offs = (const int32_t*)fa->buffers[1] + off;
          in[n.buf_main].host = (const uint8_t*)offs;
          in[n.buf_main].bytes = (uint64_t)(len + 1) * 4;
          // (device pointers: the data length lives in HBM and is not needed -- nothing is copied)
          const uint64_t dbytes = fa->buffers[2] ? (device_ptrs ? 1 : (uint64_t)offs[len]) : 0;
          in[n.buf_data].host = dbytes ? (const uint8_t*)fa->buffers[2] : nullptr;
          in[n.buf_data].bytes = dbytes;
          strs[id].offsets = offs;
          strs[id].data = (const uint8_t*)fa->buffers[2];
        } else if (len > 0) {
          throw EncodeError("fast_encode: arrow array downcast failed");
        }
        if (n.nullable) validity(n.buf_validity, fa, off, len);
        return;
      }
      case rh::NK_RECORD: {
        if (fmt != "+s") throw EncodeError("fast_encode: expected StructArray for record");
        for (size_t i = 0; i < t->fields.size(); i++) {
          int64_t hit = -1;
          for (int64_t c = 0; c < fs->n_children; c++)
            if (fs->children[c]->name && t->fields[i].name == fs->children[c]->name) { hit = c; break; }
          if (hit < 0) {
            std::string avail;
            for (int64_t c = 0; c < fs->n_children; c++) {
              if (c) avail += ", ";
              avail += "\"" + std::string(fs->children[c]->name ? fs->children[c]->name : "") + "\"";
            }
            throw EncodeError("Arrow struct missing column '" + t->fields[i].name +
                              "' required by Avro schema. Available columns: [" + avail + "]");
          }
          const ArrowArray* ca = fa->children[hit];
          bind(*t->fields[i].type, n.children[i], fs->children[hit], ca, off + ca->offset, len);
        }
        if (n.nullable) validity(n.buf_validity, fa, off, len);
        return;
      }
      case rh::NK_UNION: {
        if (fmt.rfind("+us:", 0) != 0) {
          if (fmt.rfind("+ud:", 0) == 0) throw EncodeError("fast_encode: dense unions are not supported (sparse union expected)");
          throw EncodeError("fast_encode: expected UnionArray for multi-variant union");
        }
        const int tb = fa->n_buffers == 1 ? 0 : 1;      // current C data interface: type ids only; older producers put a validity slot first
        in[n.buf_main].host = len ? (const uint8_t*)fa->buffers[tb] + off : nullptr;
        in[n.buf_main].bytes = (uint64_t)len;
        if ((size_t)fa->n_children < t->variants.size()) throw EncodeError("fast_encode: expected UnionArray for multi-variant union");
        for (size_t i = 0; i < t->variants.size(); i++) {
          const ArrowArray* ca = fa->children[i];       // schema_translate emits type ids 0..N-1 in variant order
          bind(*t->variants[i], n.children[i], fs->children[i], ca, off + ca->offset, len);
        }
        return;
      }
      case rh::NK_LIST:
      case rh::NK_MAP: {
        const bool is_map = n.kind == rh::NK_MAP;
        if (fmt != (is_map ? "+m" : "+l") || fa->n_buffers < 2 || fa->n_children < 1)
          throw EncodeError(is_map ? "fast_encode: expected MapArray for map schema" : "fast_encode: expected ListArray for array schema");
        if (fa->buffers[1]) {
          in[n.buf_main].host = (const uint8_t*)((const int32_t*)fa->buffers[1] + off);
          in[n.buf_main].bytes = (uint64_t)(len + 1) * 4;
        } else if (len > 0) {
          throw EncodeError(is_map ? "fast_encode: expected MapArray for map schema" : "fast_encode: expected ListArray for array schema");
        }
        if (n.nullable) validity(n.buf_validity, fa, off, len);
        const ArrowArray* ca = fa->children[0];
        const ArrowSchema* csch = fs->children[0];
        if (is_map) {
          if (ca->n_children < 2 || csch->n_children < 2) throw EncodeError("fast_encode: expected MapArray for map schema");
          const ArrowArray* ka = ca->children[0];
          const ArrowArray* va = ca->children[1];
          if (std::string(csch->children[0]->format ? csch->children[0]->format : "") != "u")
            throw EncodeError("fast_encode: map keys must be StringArray");
          rh::AvroType key_t;
          key_t.kind = rh::AV_STRING;
          bind(key_t, n.keys, csch->children[0], ka, ca->offset + ka->offset, ca->length);
          bind(*t->items, n.children[0], csch->children[1], va, ca->offset + va->offset, ca->length);
        } else {
          bind(*t->items, n.children[0], csch, ca, ca->offset, ca->length);
        }
        return;
      }
    }
  }
};

struct BinPriv {          // one produced BinaryArray; the k chunks share one host Slab
  const void* buffers[3];
  Slab* slab;
};
void release_binary(ArrowArray* a) {
  if (!a || !a->release) return;
  BinPriv* p = (BinPriv*)a->private_data;
  if (p->slab && p->slab->refs.fetch_sub(1) == 1) {
    p->slab->free_mem();
    delete p->slab;
  }
  delete p;
  a->release = nullptr;
}

std::string format_encode_error(const rh::ErrInfo& e, const CompiledSchema& cs, const EncodeBinder& b) {
  char buf[160];
  if (e.code == rh::EE_UNION) {
    std::snprintf(buf, sizeof buf, "fast_encode: union type_id %lld out of range", (long long)e.detail);
    return buf;
  }
  if (e.code == rh::EE_ENUM && e.pad < cs.prog.size()) {
    const int node = cs.prog[e.pad].node;
    const StrSrc& s = b.strs[node];
    std::string sym;
    if (s.offsets && s.data && b.device_ptrs) {
      int32_t o[2] = {0, 0};
      if (hipMemcpy(o, s.offsets + e.detail, sizeof o, hipMemcpyDeviceToHost) == hipSuccess && o[1] > o[0] && o[1] - o[0] < (1 << 20)) {
        sym.resize((size_t)(o[1] - o[0]));
        if (hipMemcpy(&sym[0], s.data + o[0], sym.size(), hipMemcpyDeviceToHost) != hipSuccess) sym.clear();
      }
    } else if (s.offsets && s.data) {
      sym.assign((const char*)s.data + s.offsets[e.detail], (size_t)(s.offsets[e.detail + 1] - s.offsets[e.detail]));
    }
    return "fast_encode: enum symbol '" + sym + "' not in schema";
  }
  if (e.code == rh::EE_DECIMAL) {
    std::snprintf(buf, sizeof buf, "decimal value at row %lld does not fit fixed(%u)", (long long)e.detail, e.pad);
    return buf;
  }
  if (e.code == rh::EE_DURATION) {
    std::snprintf(buf, sizeof buf, "duration value at row %lld has no Avro duration form (negative, or beyond 2^32-1 days + 2^32-1 ms)", (long long)e.detail);
    return buf;
  }
  std::snprintf(buf, sizeof buf, "encode error (code %u, op %u, detail %lld)", e.code, e.pad, (long long)e.detail);
  return buf;
}

// k BinaryArrays over one host slab holding a copy of the device output (what rh_encode returns)
void binary_chunks_to_host(const uint8_t* d_out, uint64_t out_bytes, int device, uint64_t n, uint64_t sz, uint64_t rows_last, uint32_t k,
                           const std::vector<uint64_t>& ooff, ArrowArray* out_chunks) {
  Slab* slab = slab_from_device(d_out, std::max<uint64_t>(out_bytes, 4), device);
  slab->refs.store((int)k);
  for (uint32_t c = 0; c < k; c++) {
    const uint64_t rows = n == 0 ? 0 : (c == k - 1 ? rows_last : sz);
    BinPriv* p = new BinPriv();
    p->slab = slab;
    p->buffers[0] = nullptr;
    p->buffers[1] = (const uint8_t*)slab->base + ooff[(size_t)c * 2];
    p->buffers[2] = (const uint8_t*)slab->base + ooff[(size_t)c * 2 + 1];
    ArrowArray* a = &out_chunks[c];
    a->length = (int64_t)rows; a->null_count = 0; a->offset = 0;
    a->n_buffers = 3; a->n_children = 0; a->buffers = p->buffers; a->children = nullptr; a->dictionary = nullptr;
    a->release = release_binary; a->private_data = p;
  }
}

// `dev_out` != nullptr: rh_encode_device -- the batch's buffers are device pointers, read in place, and the BinaryArrays
// stay in HBM (*dev_out owns them); else rh_encode -- host batch in, host BinaryArrays out.
int encode_impl(rh_schema* s, const ArrowArray* batch, const ArrowSchema* bschema, uint64_t num_chunks, const rh_opts* opts,
                ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, rh_device_encoded** dev_out = nullptr) {
  const bool dev = dev_out != nullptr;
  const CompiledSchema& cs = *s->cs;
  if (!cs.encode_unsupported.empty())
    throw rh::SchemaError("schema is outside the GPU encode path (" + cs.encode_unsupported +
                          ": decoded on the GPU, SURVEY 8f N4, but not encoded)");
  Timer total;
  // schema / batch mismatches are reported before any device work, like the encoder construction of
  // fast_encode.rs:33-37 that runs before the first row is written
  EncodeBinder binder(cs);
  binder.device_ptrs = dev;
  const uint64_t n = (uint64_t)batch->length;
  binder.bind(*cs.avro, 0, bschema, batch, batch->offset, (int64_t)n);

  require_device();
  int device = 0;
  if (opts && opts->device >= 0) { HIPCHK(hipSetDevice(opts->device)); device = opts->device; }
  else HIPCHK(hipGetDevice(&device));
  hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;

  // chunking of serialize.rs:15-30 (same arithmetic as the decode side)
  const uint32_t k = rh_clamp_chunks(n, num_chunks);
  const uint64_t sz = n / k, rows_last = n - (uint64_t)(k - 1) * sz;
  const uint64_t bpc64 = std::max<uint64_t>((sz + rh::kBlock - 1) / rh::kBlock, 1);
  const uint64_t nblocks64 = n == 0 ? 0 : (uint64_t)(k - 1) * bpc64 + (rows_last + rh::kBlock - 1) / rh::kBlock;
  if (nblocks64 > 0x7FFFFFFFull) throw std::invalid_argument("too many rows for one call");
  const uint32_t nblocks = (uint32_t)nblocks64;
  const int nbuf = (int)cs.bufs.size();
  const DeviceProgram& dp = device_program(s, device);

  // ---- inputs -> HBM.  Every buffer is rebased to logical row 0 and padded so that the kernels' unconditional
  // loads (encode_walk.h: row cursors up to one past the last row, 32-byte string reads) stay inside the arena; a validity bitmap the batch does
  // not carry (no nulls) is the shared all-ones bitmap at the end.
  // Device-resident input (rh_encode_device) is read where it lies; only the all-ones bitmap and one zero page for
  // absent / empty buffers are allocated.
  std::vector<uint64_t> ioff((size_t)nbuf, 0);
  uint64_t itot = 0;
  for (int b = 0; b < nbuf; b++) {
    ioff[b] = itot;
    if (!dev) itot += align_up(binder.in[b].bytes + 64, kAlign);
  }
  if (dev) itot = kAlign;                      // the zero page every absent buffer points at
  const uint64_t o_ones = itot;
  const uint64_t ones_bytes = align_up(binder.max_rows / 8 + 16, kAlign);
  itot += ones_bytes;
  Lease din(dev_pool(), itot, device);
  Timer th;
  HIPCHK(hipMemsetAsync(din.ptr() + o_ones, 0xFF, ones_bytes, stream));
  if (dev) {
    HIPCHK(hipMemsetAsync(din.ptr(), 0, kAlign, stream));
  } else {
    for (int b = 0; b < nbuf; b++) {
      if (binder.in[b].host && binder.in[b].bytes)
        HIPCHK(hipMemcpyAsync(din.ptr() + ioff[b], binder.in[b].host, binder.in[b].bytes, hipMemcpyHostToDevice, stream));
      else     // an empty column: zero offsets keep the kernels' unconditional second-level loads inside the arena
        HIPCHK(hipMemsetAsync(din.ptr() + ioff[b], 0, kAlign, stream));
    }
  }

  // ---- workspace: [first_bad][totals u64 k] | errinfo | blocksum | blockbase | in_ptr | in_bitoff | outptr
  const uint64_t o_tot = 16;
  const uint64_t ctrl_bytes = align_up(o_tot + 8ull * k, kAlign);
  const uint64_t o_err = ctrl_bytes;
  const uint64_t o_bsum = align_up(o_err + sizeof(rh::ErrInfo) * (uint64_t)nblocks, kAlign);
  const uint64_t o_bbase = align_up(o_bsum + 4ull * nblocks, kAlign);
  const uint64_t o_tab = align_up(o_bbase + 4ull * nblocks, kAlign);
  const uint64_t tab_bytes = align_up(12ull * std::max(nbuf, 1) + 16ull * k, kAlign);
  const uint64_t o_rlen = o_tab + tab_bytes;
  const uint64_t ws_bytes = o_rlen + align_up(4ull * rh::kBlock * std::max<uint64_t>(nblocks, 1), kAlign);
  Lease ws(dev_pool(), ws_bytes, device);
  Lease hctrl(pin_pool(), ctrl_bytes, device);
  Lease htab(pin_pool(), tab_bytes, device);
  HIPCHK(hipMemsetAsync(ws.ptr(), 0, ctrl_bytes, stream));
  uint64_t* h_inptr = (uint64_t*)htab.ptr();
  uint32_t* h_bitoff = (uint32_t*)(htab.ptr() + 8ull * std::max(nbuf, 1));
  void** h_out = (void**)(htab.ptr() + 12ull * std::max(nbuf, 1) + ((12ull * std::max(nbuf, 1)) % 8 ? 4 : 0));
  const uint64_t o_out = (uint64_t)((uint8_t*)h_out - htab.ptr());
  for (int b = 0; b < nbuf; b++) {
    const bool have = binder.in[b].host && binder.in[b].bytes;
    const bool bitmap = cs.bufs[b].kind == rh::BK_BITMAP;
    if (dev) h_inptr[b] = have ? (uint64_t)(uintptr_t)binder.in[b].host : (uint64_t)(uintptr_t)(din.ptr() + (bitmap ? o_ones : 0));
    else h_inptr[b] = (uint64_t)(uintptr_t)(din.ptr() + (have || !bitmap ? ioff[b] : o_ones));
    h_bitoff[b] = have ? binder.in[b].bitoff : 0;
  }

  rh::EParams E;
  std::memset(&E, 0, sizeof E);
  E.n = n; E.sz = sz; E.rows_last = rows_last; E.k = k; E.bpc = (uint32_t)bpc64; E.nblocks = nblocks;
  E.nbuf = nbuf; E.ndom = cs.ndom; E.list_depth = cs.list_depth;
  E.prog = dp.prog; E.sym_off = dp.sym_off; E.sym_data = dp.sym_data;
  E.in_ptr = (const uint64_t*)(ws.ptr() + o_tab);
  E.in_bitoff = (const uint32_t*)(ws.ptr() + o_tab + 8ull * std::max(nbuf, 1));
  E.outptr = (void* const*)(ws.ptr() + o_tab + o_out);
  E.blocksum = (uint32_t*)(ws.ptr() + o_bsum);
  E.blockbase = (const uint32_t*)(ws.ptr() + o_bbase);
  E.first_bad = (unsigned long long*)ws.ptr();
  E.errinfo = (rh::ErrInfo*)(ws.ptr() + o_err);
  E.rowlen = (uint32_t*)(ws.ptr() + o_rlen);
  // the scan kernel of the decode side, one counter
  rh::KParams SP;
  std::memset(&SP, 0, sizeof SP);
  SP.K = 1; SP.k = k; SP.bpc = (uint32_t)bpc64; SP.nblocks = nblocks;
  SP.blocksum = E.blocksum; SP.blockbase = (uint32_t*)(ws.ptr() + o_bbase); SP.totals = (uint64_t*)(ws.ptr() + o_tot);

  // kernel form: schema-specialised (hiprtc, cached per schema) or the generic interpreter, like the decode side
  const int mode = opts ? (opts->flags & 3) : RH_KERNEL_AUTO;
  const SpecKernel* sk = nullptr;
  if (mode != RH_KERNEL_GENERIC && n > 0) {
    const SpecKernel& k0 = spec_kernel(s, device, compile_policy(mode, n), true);
    if (k0.ok) sk = &k0;
    else if (mode == RH_KERNEL_SPECIALIZED) throw HipError("specialised encode kernel unavailable: " + k0.why);
  }
  const uint32_t lds = sk ? 32u : rh_enc_lds_bytes(cs.ndom, cs.list_depth);   // encode_walk.h: enc_lds_fixed_bytes
  auto launch = [&](bool emit, uint32_t lds_bytes) -> int {
    if (!sk) return emit ? rh_launch_eemit(&E, lds_bytes, stream) : rh_launch_esize(&E, lds_bytes, stream);
    rh::EParams copy = E;
    void* args[] = {&copy};
    return (int)hipModuleLaunchKernel(emit ? sk->emit_fn : sk->size_fn, nblocks, 1, 1, rh::kBlock, 1, 1, lds_bytes, stream, args, nullptr);
  };
  auto check_bad = [&](const uint8_t* h, const char* pass) {
    unsigned long long fb = *(const unsigned long long*)h;
    if (!fb) return;
    if (std::getenv("RUHVRO_HIP_DEBUG")) std::fprintf(stderr, "rh_encode: %s pass reports first_bad=%llx\n", pass, fb);
    const uint64_t rec = ~fb;
    uint64_t c = sz ? std::min<uint64_t>(rec / sz, k - 1) : 0;
    uint64_t bl = c * bpc64 + (rec - c * sz) / rh::kBlock;
    rh::ErrInfo ei;
    HIPCHK(hipMemcpy(&ei, E.errinfo + bl, sizeof ei, hipMemcpyDeviceToHost));
    throw EncodeError(format_encode_error(ei, cs, binder));
  };

  // first launch needs the input tables on the device (outptr is filled in later)
  HIPCHK(hipMemcpyAsync(ws.ptr() + o_tab, htab.ptr(), tab_bytes, hipMemcpyHostToDevice, stream));
  Events ev;
  if (stats) ev.init();
  ev.rec(0, stream);
  std::vector<uint64_t> totals((size_t)k, 0);
  if (n > 0) {
    if (launch(false, lds)) throw HipError("e_size launch failed");
    ev.rec(1, stream);
    if (rh_launch_scan(&SP, stream, nullptr, nullptr)) throw HipError("k_scan launch failed");
    ev.rec(2, stream);
    HIPCHK(hipMemcpyAsync(hctrl.ptr(), ws.ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    check_bad(hctrl.ptr(), "size");
    std::memcpy(totals.data(), hctrl.ptr() + o_tot, 8ull * k);
  } else {
    HIPCHK(hipStreamSynchronize(stream));
    ev.rec(1, stream);
    ev.rec(2, stream);
  }
  const float h2d = th.ms();
  for (auto t : totals)
    if (t > 0x7FFFFFFFull) throw EncodeError("offset overflow: a chunk's encoded bytes exceed the 2^31-1 limit of BinaryArray offsets");

  // ---- output arena: per chunk offsets i32[rows+1] + data
  std::vector<uint64_t> ooff((size_t)k * 2);
  uint64_t otot = 0, exact = 0;
  for (uint32_t c = 0; c < k; c++) {
    const uint64_t rows = n == 0 ? 0 : (c == k - 1 ? rows_last : sz);
    ooff[(size_t)c * 2] = otot;
    otot += align_up((rows + 1) * 4, kAlign);
    ooff[(size_t)c * 2 + 1] = otot;
    otot += align_up(std::max<uint64_t>(totals[c], 8), kAlign);
    exact += (rows + 1) * 4 + totals[c];
  }
  Lease dout(dev_pool(), std::max<uint64_t>(otot, kAlign), device);
  for (uint32_t c = 0; c < k; c++) {
    h_out[(size_t)c * 2] = dout.ptr() + ooff[(size_t)c * 2];
    h_out[(size_t)c * 2 + 1] = dout.ptr() + ooff[(size_t)c * 2 + 1];
  }
  HIPCHK(hipMemcpyAsync(ws.ptr() + o_tab + o_out, (uint8_t*)h_out, 16ull * k, hipMemcpyHostToDevice, stream));
  if (n == 0) HIPCHK(hipMemsetAsync(dout.ptr(), 0, 4, stream));   // offsets[0] of the single empty chunk
  // staging window of rh_e_emit: the mean workgroup's bytes + 15 % + 2 KB, within the 64 KB a launch gets by default
  uint64_t sum = 0;
  for (auto t : totals) sum += t;
  uint64_t win = nblocks ? sum / nblocks : 0;
  win = align_up(win + win * 15 / 100 + 2048, 16);
  win = std::min<uint64_t>(win, (65536 - lds) & ~15ull);
  // string staging areas of the specialised kernel (encode_walk.h e_string_cofetch), behind the window; the window gives
  // up slack rather than the 4-workgroups-per-CU occupancy when the mean workgroup still fits with ~3 % + 512 bytes
  uint32_t stage = 0;
  if (sk) {
    stage = 4 * rh::kStageStride;
    const uint64_t mean = nblocks ? sum / nblocks : 0;
    const uint64_t cap4 = (40960 - lds - stage) & ~15ull;
    if (win + stage + lds > 40960 && mean + mean * 3 / 100 + 512 <= cap4) win = cap4;
    win = std::min<uint64_t>(win, (65536 - lds - stage) & ~15ull);
  }
  E.win_bytes = (uint32_t)win;
  E.stage_bytes = stage;
  static const bool profile = [] { const char* e = std::getenv("RUHVRO_HIP_PROFILE"); return e && *e && *e != '0'; }();
  Lease prof_buf;
  if (profile && sk) {
    prof_buf = Lease(dev_pool(), 64 * 32 * 8, device);
    HIPCHK(hipMemsetAsync(prof_buf.ptr(), 0, 64 * 32 * 8, stream));
    E.prof = (unsigned long long*)prof_buf.ptr();
  }
  ev.rec(3, stream);
  if (n > 0 && launch(true, lds + E.win_bytes + E.stage_bytes)) throw HipError("e_emit launch failed");
  ev.rec(4, stream);
  HIPCHK(hipMemcpyAsync(hctrl.ptr(), ws.ptr(), 16, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  check_bad(hctrl.ptr(), "emit");
  if (profile && sk) {     // phase cycles of rh_espec_emit (encode_walk.h PhaseClock): 0 prologue, 1 offsets, 2..19 walk stretches, 20..22 tail
    unsigned long long hr[64 * 32], h[32] = {0};
    HIPCHK(hipMemcpy(hr, prof_buf.ptr(), sizeof hr, hipMemcpyDeviceToHost));
    for (int r0 = 0; r0 < 64; r0++)
      for (int i = 0; i < 32; i++) h[i] += hr[r0 * 32 + i];
    const double waves = (double)nblocks * 4;
    std::fprintf(stderr, "[ruhvro_hip profile] e_emit cycles/wave: rowlen+scan+barrier=%.0f offsets=%.0f | walk:", h[0] / waves, h[1] / waves);
    for (int i = 2; i < 20; i++)
      if (h[i]) std::fprintf(stderr, " [%d]=%.0f", i, h[i] / waves);
    std::fprintf(stderr, " | walk_tail=%.0f barrier=%.0f stream_out=%.0f\n", h[20] / waves, h[21] / waves, h[22] / waves);
  }

  // ---- results: left in HBM (rh_encode_device) or -> host, one slab shared by the k BinaryArrays
  Timer td;
  if (dev) {
    auto res = std::make_unique<rh_device_encoded>();
    res->device = device; res->n = n; res->sz = sz; res->rows_last = rows_last; res->k = k;
    res->out = std::move(dout);
    res->out_bytes = std::max<uint64_t>(otot, 4);
    res->ooff = ooff;
    res->data_bytes = totals;
    res->exact = exact;
    *dev_out = res.release();
  } else {
    binary_chunks_to_host(dout.ptr(), otot, device, n, sz, rows_last, k, ooff, out_chunks);
  }
  if (out_k) *out_k = k;
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->records = n;
    stats->output_bytes = exact;
    for (int b = 0; b < nbuf; b++) stats->input_bytes += binder.in[b].bytes;   // (device input: string data bytes are not known here)
    stats->chunks = k;
    stats->blocks = nblocks;
    stats->h2d_ms = h2d;
    stats->size_kernel_ms = ev.ms(0, 1);
    stats->scan_kernel_ms = ev.ms(1, 2);
    stats->emit_kernel_ms = ev.ms(3, 4);
    stats->d2h_ms = td.ms();
    stats->total_ms = total.ms();
    stats->specialized = sk ? 1 : 0;
    stats->lds_bytes = lds + E.win_bytes;
  }
  return RH_OK;
}

}  // namespace

extern "C" int rh_encode_device(const rh_schema* s, const struct ArrowArray* batch, const struct ArrowSchema* batch_schema, uint64_t num_chunks,
                                const rh_opts* opts, rh_device_encoded** out, rh_stats* stats, char** err) {
  if (!s || !batch || !batch_schema || !out) return RH_ERR_ARGUMENT;
  *out = nullptr;
  return guarded(err, [&] { return encode_impl(const_cast<rh_schema*>(s), batch, batch_schema, num_chunks, opts, nullptr, nullptr, stats, out); });
}
extern "C" uint32_t rh_device_encoded_chunks(const rh_device_encoded* r) { return r ? r->k : 0; }
extern "C" uint64_t rh_device_encoded_output_bytes(const rh_device_encoded* r) { return r ? r->exact : 0; }
extern "C" int rh_device_encoded_export(rh_device_encoded* r, uint32_t chunk, struct ArrowDeviceArray* out) {
  if (!r || !out || chunk >= r->k) return RH_ERR_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  BinPriv* p = new BinPriv();
  p->slab = nullptr;                     // a view: the memory belongs to the rh_device_encoded
  p->buffers[0] = nullptr;
  p->buffers[1] = r->out.ptr() + r->ooff[(size_t)chunk * 2];
  p->buffers[2] = r->out.ptr() + r->ooff[(size_t)chunk * 2 + 1];
  ArrowArray* a = &out->array;
  a->length = (int64_t)r->rows(chunk); a->null_count = 0; a->offset = 0;
  a->n_buffers = 3; a->n_children = 0; a->buffers = p->buffers; a->children = nullptr; a->dictionary = nullptr;
  a->release = release_binary; a->private_data = p;
  out->device_id = r->device;
  out->device_type = ARROW_DEVICE_ROCM;
  out->sync_event = nullptr;             // the producing stream was synchronised before the result was returned
  return RH_OK;
}
extern "C" int rh_device_encoded_to_host(rh_device_encoded* r, struct ArrowArray* out_chunks, char** err) {
  if (!r || !out_chunks) return RH_ERR_ARGUMENT;
  std::memset(out_chunks, 0, sizeof(ArrowArray) * r->k);
  return guarded(err, [&] {
    HIPCHK(hipSetDevice(r->device));
    binary_chunks_to_host(r->out.ptr(), r->out_bytes, r->device, r->n, r->sz, r->rows_last, r->k, r->ooff, out_chunks);
    return (int)RH_OK;
  });
}
extern "C" void rh_device_encoded_free(rh_device_encoded* r) { delete r; }

extern "C" int rh_encode(const rh_schema* s, const struct ArrowArray* batch, const struct ArrowSchema* batch_schema, uint64_t num_chunks,
                         const rh_opts* opts, struct ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, char** err) {
  if (!s || !batch || !batch_schema || !out_chunks) return RH_ERR_ARGUMENT;
  // the caller only "provides room": zero it so that every failure path can tell produced chunks from garbage
  std::memset(out_chunks, 0, sizeof(ArrowArray) * rh_clamp_chunks(batch->length < 0 ? 0 : (uint64_t)batch->length, num_chunks));
  return guarded(err, [&] { return encode_impl(const_cast<rh_schema*>(s), batch, batch_schema, num_chunks, opts, out_chunks, out_k, stats); });
}

