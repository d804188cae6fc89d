// Arrow C Data / C Device Data export of the produced buffers and the host copies of results.
#include "engine_internal.h"

using namespace rhe;

void rh_device_result::fill_tables() {
  const CompiledSchema& c_s = *cs;
  const int nbuf = (int)c_s.bufs.size();
  dom_rows.assign((size_t)c_s.ndom * k, 0);
  for (uint32_t c = 0; c < k; c++) {
    dom_rows[c] = n == 0 ? 0 : (c == k - 1 ? rows_last : sz);
    for (int d = 1; d < c_s.ndom; d++) dom_rows[(size_t)d * k + c] = data_bytes[(size_t)(d - 1) * k + c];
  }
  buf_off.assign((size_t)nbuf * k, 0);
  buf_size.assign((size_t)nbuf * k, 0);
  uint64_t off = 0, exact = 0;
  const bool capl = !layout_bytes.empty();      // slots as the single-pass form reserved them; sizes are the real ones
  for (uint32_t c = 0; c < k; c++) {
    for (int b = 0; b < nbuf; b++) {
      const rh::BufDesc& d = c_s.bufs[b];
      uint64_t ex = 0;
      const uint64_t bytes = rh::buf_bytes(d.kind, rows(d.dom, c), d.kind == rh::BK_DATA ? data_bytes[(size_t)d.counter * k + c] : 0, &ex,
                                           (uint32_t)d.counter);
      uint64_t slot = bytes;
      if (capl) {
        const uint64_t crow = d.dom == 0 ? rows(0, c) : layout_bytes[(size_t)(d.dom - 1) * k + c];
        slot = rh::buf_bytes(d.kind, crow, d.kind == rh::BK_DATA ? layout_bytes[(size_t)d.counter * k + c] : 0, nullptr, (uint32_t)d.counter);
      }
      buf_off[(size_t)b * k + c] = off;
      buf_size[(size_t)b * k + c] = bytes;
      off += rh::buf_slot_bytes(slot);
      exact += ex;
    }
  }
  arena_bytes = std::max<uint64_t>(off, 256);
  output_bytes = exact;
  tables_done = true;
}


namespace rhe {

struct ArrayPriv {
  std::vector<const void*> buffers;
  std::vector<ArrowArray*> children;
  Slab* slab = nullptr;   // top-level arrays only
};

void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  ArrayPriv* p = (ArrayPriv*)a->private_data;
  for (ArrowArray* c : p->children) {
    if (c->release) c->release(c);
    delete c;
  }
  if (p->slab && p->slab->refs.fetch_sub(1) == 1) {
    p->slab->free_mem();
    delete p->slab;
  }
  delete p;
  a->release = nullptr;
}

void init_array(ArrowArray* a, int64_t length, int64_t null_count, std::vector<const void*> bufs,
                std::vector<ArrowArray*> kids) {
  ArrayPriv* p = new ArrayPriv();
  p->buffers = std::move(bufs);
  p->children = std::move(kids);
  a->length = length;
  a->null_count = null_count;
  a->offset = 0;
  a->n_buffers = (int64_t)p->buffers.size();
  a->n_children = (int64_t)p->children.size();
  a->buffers = p->buffers.empty() ? nullptr : p->buffers.data();
  a->children = p->children.empty() ? nullptr : p->children.data();
  a->dictionary = nullptr;
  a->release = release_array;
  a->private_data = p;
}

// Builds the array of decoder node `id` for chunk c; `base` is the arena base (host slab or device).
ArrowArray* export_node(const rh_device_result& r, int id, uint32_t c, const uint8_t* base) {
  const CompiledSchema& cs = *r.cs;
  const DecNode& n = cs.nodes[id];
  const int64_t len = (int64_t)r.rows(n.dom, c);
  const int64_t nulls = (int64_t)r.nullcount[(size_t)id * r.k + c];
  auto bp = [&](int buf) -> const void* { return buf < 0 ? nullptr : base + r.buf_off[(size_t)buf * r.k + c]; };
  ArrowArray* a = new ArrowArray();
  switch (n.kind) {
    case rh::NK_FIXED:
      // leaf builders keep a lazy null buffer: bitmap only if a null was appended
      init_array(a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, {});
      break;
    case rh::NK_STRING: case rh::NK_ENUM:
      init_array(a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main), bp(n.buf_data)}, {});
      break;
    case rh::NK_BIN:        // FixedSizeBinary / Decimal128: lazy validity like every leaf builder, one values buffer
      init_array(a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, {});
      break;
    case rh::NK_NULL:
      init_array(a, len, len, {}, {});
      break;
    case rh::NK_RECORD: {   // fast_decode.rs:618-639: validity iff the record decoder is nullable
      std::vector<ArrowArray*> kids;
      for (int ch : n.children) kids.push_back(export_node(r, ch, c, base));
      init_array(a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr}, std::move(kids));
      break;
    }
    case rh::NK_UNION: {    // fast_decode.rs:670-683: sparse, type_ids only
      std::vector<ArrowArray*> kids;
      for (int ch : n.children) kids.push_back(export_node(r, ch, c, base));
      init_array(a, len, 0, {bp(n.buf_main)}, std::move(kids));
      break;
    }
    case rh::NK_LIST: {     // fast_decode.rs:729-741
      ArrowArray* item = export_node(r, n.children[0], c, base);
      init_array(a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, {item});
      break;
    }
    case rh::NK_MAP: {      // fast_decode.rs:772-798
      ArrowArray* keys = export_node(r, n.keys, c, base);
      ArrowArray* vals = export_node(r, n.children[0], c, base);
      ArrowArray* entries = new ArrowArray();
      init_array(entries, (int64_t)r.rows(n.child_dom, c), 0, {nullptr}, {keys, vals});
      init_array(a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, {entries});
      break;
    }
  }
  return a;
}

void export_chunk(const rh_device_result& r, uint32_t c, const uint8_t* base, Slab* slab, ArrowArray* out) {
  const DecNode& top = r.cs->nodes[0];
  std::vector<ArrowArray*> kids;
  for (int ch : top.children) kids.push_back(export_node(r, ch, c, base));
  init_array(out, (int64_t)r.rows(0, c), 0, {nullptr}, std::move(kids));
  if (slab) {
    ((ArrayPriv*)out->private_data)->slab = slab;
    slab->refs.fetch_add(1);
  }
}

// ---- ArrowSchema export -----------------------------------------------------
struct SchemaPriv {
  std::string format, name, metadata;
  std::vector<ArrowSchema*> children;
};

void release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  SchemaPriv* p = (SchemaPriv*)s->private_data;
  for (ArrowSchema* c : p->children) {
    if (c->release) c->release(c);
    delete c;
  }
  delete p;
  s->release = nullptr;
}

void export_field(const rh::ArrowField& f, ArrowSchema* out) {
  SchemaPriv* p = new SchemaPriv();
  p->format = f.format;
  p->name = f.name;
  if (!f.metadata.empty()) {   // int32 count, then (int32 len, bytes) x2 per pair, native endianness
    auto put32 = [&](int32_t v) { p->metadata.append((const char*)&v, 4); };
    put32((int32_t)f.metadata.size());
    for (auto& kv : f.metadata) {
      put32((int32_t)kv.first.size()); p->metadata += kv.first;
      put32((int32_t)kv.second.size()); p->metadata += kv.second;
    }
  }
  for (auto& ch : f.children) {
    ArrowSchema* cs = new ArrowSchema();
    export_field(ch, cs);
    p->children.push_back(cs);
  }
  out->format = p->format.c_str();
  out->name = p->name.c_str();
  out->metadata = p->metadata.empty() ? nullptr : p->metadata.data();
  out->flags = (f.nullable ? ARROW_FLAG_NULLABLE : 0) | (f.map_keys_sorted ? ARROW_FLAG_MAP_KEYS_SORTED : 0);
  out->n_children = (int64_t)p->children.size();
  out->children = p->children.empty() ? nullptr : p->children.data();
  out->dictionary = nullptr;
  out->release = release_schema;
  out->private_data = p;
}


// Device range -> freshly owned host memory.  Large results land in pooled PINNED memory when a block is idle in the pool (the
// copy then runs at PCIe speed, 57 GB/s measured), else in pageable memory (15-22 GB/s) while the pool is refilled in the
// background: hipHostMalloc costs 0.2 ms per MB (tools/d2hcost.hip) -- on the call's own path that was 32 ms for the results
// of a 1M-record call whose predecessor's batches were still alive, twice the call itself.
Slab* slab_from_device(const uint8_t* dptr, uint64_t bytes, int device, hipStream_t stream) {
  Slab* slab = new Slab();
  try {
    if (bytes >= (1ull << 20)) {
      slab->pinned = pin_pool().try_get(bytes, device);
      if (slab->pinned.p) {
        Slab::pinned_result_bytes().fetch_add(slab->pinned.size);
        slab->base = slab->pinned.p;
      } else {
        pin_pool().prefetch(bytes + bytes / 16, device, pinned_budget_left());
      }
    }
    if (!slab->base && posix_memalign(&slab->base, 64, std::max<uint64_t>(bytes, 64)) != 0) throw std::bad_alloc();
    if (bytes) {
      hipError_t e = hipMemcpyAsync(slab->base, dptr, bytes, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) throw HipError(std::string("D2H copy failed: ") + hipGetErrorString(e));
    }
  } catch (...) {
    slab->free_mem();
    delete slab;
    throw;
  }
  return slab;
}

int to_host_impl(rh_device_result* r, ArrowArray* out_chunks, hipStream_t stream) {
  settle(r);
  if (!r->parts.empty()) {
    uint32_t built = 0;
    try {
      for (size_t g = 0; g < r->parts.size(); g++) {
        to_host_impl(r->parts[g].get(), out_chunks + r->part_chunk0[g], stream);
        built = r->part_chunk0[g] + r->parts[g]->k;
      }
    } catch (...) {
      for (uint32_t c = 0; c < built; c++)
        if (out_chunks[c].release) out_chunks[c].release(&out_chunks[c]);
      throw;
    }
    return 0;
  }
  r->tables();
  Slab* slab = nullptr;
  if (r->arena_host) {          // the emit kernel wrote into pinned host memory: the result takes the block over as it is
    slab = new Slab();
    slab->pinned = r->arena.b;
    slab->base = r->arena.b.p;
    Slab::pinned_result_bytes().fetch_add(slab->pinned.size);
    r->arena.pool = nullptr;
    r->arena.b = Block();
    r->arena_host = false;
  } else {
    slab = slab_from_device(r->arena.ptr(), r->arena_bytes, r->device, stream);
  }
  slab->refs.store(1);   // guard while building
  uint32_t built = 0;
  try {
    for (; built < r->k; built++) export_chunk(*r, built, (const uint8_t*)slab->base, slab, &out_chunks[built]);
  } catch (...) {          // drop the chunks already exported (each holds a slab reference), then the guard
    for (uint32_t c = 0; c < built; c++)
      if (out_chunks[c].release) out_chunks[c].release(&out_chunks[c]);
    if (slab->refs.fetch_sub(1) == 1) { slab->free_mem(); delete slab; }
    throw;
  }
  if (slab->refs.fetch_sub(1) == 1) { slab->free_mem(); delete slab; }
  return 0;
}


}  // namespace rhe
