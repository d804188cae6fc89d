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

// One allocation per exported chunk holds every ArrowArray of its tree below the top-level one (which lives in the caller's
// memory), their buffer-pointer arrays and their child-pointer arrays: a chunk of the benchmark schema is ~40 arrays, and one
// `new` + two vectors per array were ~1,200 allocations per 8-chunk call (23 us of a 0.27 ms call of 10,000 records, and most of
// what a call with num_chunks = n spends on the host).  Every array of the tree holds one reference on the arena, so a consumer
// may move a child out and release it after its parent (Arrow C Data interface: a moved child is released on its own).
struct ChunkArena {
  std::atomic<int> refs{0};
  Slab* slab = nullptr;             // the host copy the buffers point into (nullptr: device memory owned by the result)
  ArrowArray* nodes = nullptr;      // storage, carved from the same allocation
  const void** bufs = nullptr;
  ArrowArray** kids = nullptr;
  size_t n_nodes = 0, n_bufs = 0, n_kids = 0;      // used so far
};

void release_node(ArrowArray* a) {
  if (!a || !a->release) return;
  ChunkArena* ar = (ChunkArena*)a->private_data;
  for (int64_t i = 0; i < a->n_children; i++) {
    ArrowArray* c = a->children[i];
    if (c && c->release) c->release(c);
  }
  a->release = nullptr;
  if (ar->refs.fetch_sub(1) == 1) {
    if (ar->slab && ar->slab->refs.fetch_sub(1) == 1) {
      ar->slab->free_mem();
      delete ar->slab;
    }
    ar->~ChunkArena();
    std::free(ar);
  }
}

// arrays / buffer slots / child slots of the tree under node `id` (the shape export_node builds)
void count_node(const CompiledSchema& cs, int id, size_t& nn, size_t& nb, size_t& nk) {
  const DecNode& n = cs.nodes[id];
  nn += 1;
  switch (n.kind) {
    case rh::NK_FIXED: case rh::NK_BIN: nb += 2; break;
    case rh::NK_STRING: case rh::NK_ENUM: nb += 3; break;
    case rh::NK_NULL: break;
    case rh::NK_RECORD:
      nb += 1; nk += n.children.size();
      for (int ch : n.children) count_node(cs, ch, nn, nb, nk);
      break;
    case rh::NK_UNION:
      nb += 1; nk += n.children.size();
      for (int ch : n.children) count_node(cs, ch, nn, nb, nk);
      break;
    case rh::NK_LIST:
      nb += 2; nk += 1;
      count_node(cs, n.children[0], nn, nb, nk);
      break;
    case rh::NK_MAP:
      nb += 2 + 1; nk += 1 + 2; nn += 1;          // the map, its entries struct, keys + values
      count_node(cs, n.keys, nn, nb, nk);
      count_node(cs, n.children[0], nn, nb, nk);
      break;
  }
}

void init_array(ChunkArena* ar, ArrowArray* a, int64_t length, int64_t null_count, std::initializer_list<const void*> bufs, size_t nkids) {
  a->length = length;
  a->null_count = null_count;
  a->offset = 0;
  a->n_buffers = (int64_t)bufs.size();
  a->n_children = (int64_t)nkids;
  a->buffers = bufs.size() ? ar->bufs + ar->n_bufs : nullptr;
  for (const void* bptr : bufs) ar->bufs[ar->n_bufs++] = bptr;
  a->children = nkids ? ar->kids + ar->n_kids : nullptr;
  ar->n_kids += nkids;                              // (the caller fills a->children[0 .. nkids))
  a->dictionary = nullptr;
  a->release = release_node;
  a->private_data = ar;
  ar->refs.fetch_add(1, std::memory_order_relaxed);
}

// Builds the array of decoder node `id` for chunk c; `base` is the arena base (host slab or device).
ArrowArray* export_node(ChunkArena* ar, const rh_device_result& r, int id, uint32_t c, const uint8_t* base) {
  const CompiledSchema& cs = *r.cs;
  const DecNode& n = cs.nodes[id];
  const int64_t len = (int64_t)r.rows(n.dom, c);
  const int64_t nulls = (int64_t)r.nullcount[(size_t)id * r.k + c];
  auto bp = [&](int buf) -> const void* { return buf < 0 ? nullptr : base + r.buf_off[(size_t)buf * r.k + c]; };
  ArrowArray* a = &ar->nodes[ar->n_nodes++];
  switch (n.kind) {
    case rh::NK_FIXED:
      // leaf builders keep a lazy null buffer: bitmap only if a null was appended
      init_array(ar, a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, 0);
      break;
    case rh::NK_STRING: case rh::NK_ENUM:
      init_array(ar, a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main), bp(n.buf_data)}, 0);
      break;
    case rh::NK_BIN:        // FixedSizeBinary / Decimal128: lazy validity like every leaf builder, one values buffer
      init_array(ar, a, len, nulls, {nulls > 0 ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, 0);
      break;
    case rh::NK_NULL:
      init_array(ar, a, len, len, {}, 0);
      break;
    case rh::NK_RECORD: {   // fast_decode.rs:618-639: validity iff the record decoder is nullable
      init_array(ar, a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr}, n.children.size());
      for (size_t i = 0; i < n.children.size(); i++) a->children[i] = export_node(ar, r, n.children[i], c, base);
      break;
    }
    case rh::NK_UNION: {    // fast_decode.rs:670-683: sparse, type_ids only
      init_array(ar, a, len, 0, {bp(n.buf_main)}, n.children.size());
      for (size_t i = 0; i < n.children.size(); i++) a->children[i] = export_node(ar, r, n.children[i], c, base);
      break;
    }
    case rh::NK_LIST: {     // fast_decode.rs:729-741
      init_array(ar, a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, 1);
      a->children[0] = export_node(ar, r, n.children[0], c, base);
      break;
    }
    case rh::NK_MAP: {      // fast_decode.rs:772-798
      init_array(ar, a, len, n.nullable ? nulls : 0, {n.nullable ? bp(n.buf_validity) : nullptr, bp(n.buf_main)}, 1);
      ArrowArray* entries = &ar->nodes[ar->n_nodes++];
      init_array(ar, entries, (int64_t)r.rows(n.child_dom, c), 0, {nullptr}, 2);
      entries->children[0] = export_node(ar, r, n.keys, c, base);
      entries->children[1] = export_node(ar, r, n.children[0], c, base);
      a->children[0] = entries;
      break;
    }
  }
  return a;
}

void export_chunk(const rh_device_result& r, uint32_t c, const uint8_t* base, Slab* slab, ArrowArray* out) {
  const CompiledSchema& cs = *r.cs;
  const DecNode& top = cs.nodes[0];
  size_t nn = 0, nb = 1, nk = top.children.size();
  for (int ch : top.children) count_node(cs, ch, nn, nb, nk);
  const size_t o_nodes = (sizeof(ChunkArena) + 15) & ~(size_t)15;
  const size_t o_bufs = o_nodes + nn * sizeof(ArrowArray), o_kids = o_bufs + nb * sizeof(void*);
  void* mem = std::malloc(o_kids + nk * sizeof(void*) + 8);
  if (!mem) throw std::bad_alloc();
  ChunkArena* ar = new (mem) ChunkArena();
  ar->nodes = reinterpret_cast<ArrowArray*>((uint8_t*)mem + o_nodes);
  ar->bufs = reinterpret_cast<const void**>((uint8_t*)mem + o_bufs);
  ar->kids = reinterpret_cast<ArrowArray**>((uint8_t*)mem + o_kids);
  init_array(ar, out, (int64_t)r.rows(0, c), 0, {nullptr}, top.children.size());
  for (size_t i = 0; i < top.children.size(); i++) out->children[i] = export_node(ar, r, top.children[i], c, base);
  if (slab) {
    ar->slab = slab;
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
