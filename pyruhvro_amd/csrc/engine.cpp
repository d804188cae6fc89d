// C ABI of the direct-decode engine (include/ruhvro_hip.h): schema cache objects,
// device memory pools, the k_size -> k_scan -> k_emit launch sequence, and the
// Arrow C Data / C Device Data export of the produced buffers.
//
// Replaces, for the one hot path, the reference's chunk driver
// (ruhvro/src/deserialize.rs:76-121: pack, slice, one task per chunk, ordered
// join) -- the device boundary takes the place of the spawn_blocking boundary.
// There is NO CPU decode fallback in this library: without a HIP device every
// decode entry point fails with RH_ERR_RUNTIME.
#include "engine_internal.h"

#include <sched.h>

using namespace rhe;

// ===========================================================================
// C ABI
// ===========================================================================
// RUHVRO_HIP_STATS=1: one JSON line per host decode call on stderr (the reference has no metrics
// at all; this is the per-call stats struct of the C ABI, printed) -- entry point, records, bytes, per-stage ms, per-shard
// when the call was dealt to several devices.
bool stats_line_on() {
  static const bool on = [] { const char* e = std::getenv("RUHVRO_HIP_STATS"); return e && *e && *e != '0'; }();
  return on;
}
void stats_json(std::string& o, const rh_stats& st) {
  char buf[512];
  std::snprintf(buf, sizeof buf,
                "{\"records\": %llu, \"input_bytes\": %llu, \"output_bytes\": %llu, \"chunks\": %u, \"blocks\": %u, \"pack_ms\": %.4f, "
                "\"h2d_ms\": %.4f, \"size_kernel_ms\": %.4f, \"scan_kernel_ms\": %.4f, \"emit_kernel_ms\": %.4f, \"d2h_ms\": %.4f, "
                "\"total_ms\": %.4f, \"specialized\": %u, \"lds_bytes\": %u}",
                (unsigned long long)st.records, (unsigned long long)st.input_bytes, (unsigned long long)st.output_bytes, st.chunks,
                st.blocks, st.pack_ms, st.h2d_ms, st.size_kernel_ms, st.scan_kernel_ms, st.emit_kernel_ms, st.d2h_ms, st.total_ms,
                st.specialized, st.lds_bytes);
  o += buf;
}
void stats_line(const char* entry, int rc, const rh_stats& st, const rh_opts* opts, const rh_stats* shards) {
  std::string o = "{\"ruhvro_hip\": \"";
  o += entry;
  o += "\", \"rc\": " + std::to_string(rc) + ", \"stats\": ";
  stats_json(o, st);
  if (shards && opts && opts->n_devices > 1) {
    o += ", \"devices\": [";
    for (uint32_t i = 0; i < opts->n_devices; i++) {
      if (i) o += ", ";
      o += "{\"device\": " + std::to_string(opts->devices[i]) + ", \"stats\": ";
      stats_json(o, shards[i]);
      o += "}";
    }
    o += "]";
  }
  o += "}\n";
  std::fputs(o.c_str(), stderr);
}
// runs a host decode entry point with the stats struct forced on when the line was asked for
template <class F>
int with_stats_line(const char* entry, const rh_opts* opts, rh_stats* stats, F&& f) {
  if (!stats_line_on()) return f(opts, stats);
  rh_stats local;
  std::memset(&local, 0, sizeof local);
  rh_stats* st = stats ? stats : &local;
  std::vector<rh_stats> shard_st;
  rh_opts o2;
  const rh_opts* use = opts;
  if (opts && opts->n_devices > 1 && !opts->device_stats) {       // per-shard timings for the line
    shard_st.resize(opts->n_devices);
    o2 = *opts;
    o2.device_stats = shard_st.data();
    use = &o2;
  }
  const int rc = f(use, st);
  stats_line(entry, rc, *st, use, use ? use->device_stats : nullptr);
  return rc;
}

extern "C" {

int rh_abi_version(void) { return RH_ABI_VERSION; }
uint32_t rh_effective_cpus(void) { return effective_cpus(); }

uint32_t rh_engine_counters(uint64_t* out, uint32_t n) {
  for (uint32_t i = 0; i < n && i < (uint32_t)RH_CTR_COUNT; i++) out[i] = g_counters[i].load(std::memory_order_relaxed);
  return RH_CTR_COUNT;
}

int rh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Test / measurement hook (not part of the drop-in surface): `shards` host threads gather their contiguous share of the n
// record slices at the same time, each with `threads_per_shard` helpers, exactly as the shards of a multi-GPU rh_decode
// call do (gather_slices) -- into pageable memory (pinned = 0; needs no GPU) or pinned memory (pinned = 1).  Destinations
// are allocated and touched before the clock starts.  Writes the best wall time of `reps` rounds to *best_ms and returns
// the payload bytes gathered per round (0 on failure).
uint64_t rh_bench_gather(const uint8_t* const* ptrs, const uint64_t* lens, uint64_t n, uint32_t shards, uint32_t threads_per_shard,
                         int pinned, uint32_t reps, double* best_ms) {
  if (!ptrs || !lens || !best_ms || shards == 0) return 0;
  // pinned: 0 = pageable destinations, 1 = hipHostMalloc, 2 = pageable and NUMA-PLACED: shard j's host thread (and its helpers,
  // which inherit the mask) is bound to the cpus of node j mod <nodes> and allocates + first-touches its destination there
  const std::vector<std::vector<int>> nodes = pinned == 2 ? numa_node_cpus() : std::vector<std::vector<int>>();      // (engine_pools.cpp)
  struct Dst { uint8_t* p = nullptr; uint64_t bytes = 0, rows0 = 0, rows = 0, o_off = 0; };
  std::vector<Dst> dst(shards);
  uint64_t total = 0;
  bool ok = true;
  for (uint32_t j = 0; j < shards; j++) {
    Dst& d = dst[j];
    d.rows0 = n * j / shards; d.rows = n * (j + 1) / shards - d.rows0;
    uint64_t b = 0;
    for (uint64_t i = 0; i < d.rows; i++) b += lens[d.rows0 + i];
    total += b;
    d.o_off = align_up(16 + b + 32, kAlign);
    d.bytes = d.o_off + 8 * (d.rows + 1);
    if (pinned == 1) ok = ok && hipHostMalloc((void**)&d.p, d.bytes, hipHostMallocDefault) == hipSuccess;
    else ok = ok && posix_memalign((void**)&d.p, 4096, d.bytes) == 0;
    if (ok && pinned != 2) std::memset(d.p, 0, d.bytes);
  }
  auto bind = [&](unsigned j) {
    if (nodes.size() < 2) return;
    const std::vector<int>& cpus = nodes[j % nodes.size()];
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof set, &set);
  };
  if (ok && pinned == 2)
    run_threads(shards, [&](unsigned j) { bind(j); std::memset(dst[j].p, 0, dst[j].bytes); });     // first touch on the shard's node
  double best = 1e30;
  for (uint32_t r = 0; ok && r < std::max(reps, 1u); r++) {
    Timer t;
    run_threads(shards, [&](unsigned j) {
      Dst& d = dst[j];
      if (pinned == 2) bind(j);
      gather_into(ptrs + d.rows0, lens + d.rows0, d.rows, threads_per_shard, d.p + 16, (uint64_t*)(d.p + d.o_off));
    });
    best = std::min(best, (double)t.ms());
  }
  for (Dst& d : dst)
    if (d.p) { if (pinned == 1) (void)hipHostFree(d.p); else std::free(d.p); }
  *best_ms = best;
  return ok ? total : 0;
}

uint32_t rh_numa_nodes(void) { return (uint32_t)numa_node_cpus().size(); }

int rh_current_device(void) {
  int n = 0, d = -1;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -1;
  if (hipGetDevice(&d) != hipSuccess) return -1;
  return d;
}

void rh_free_string(char* s) { std::free(s); }

uint32_t rh_clamp_chunks(uint64_t n, uint64_t num_chunks) {   // deserialize.rs:53-55
  uint64_t k = std::max<uint64_t>(num_chunks, 1);
  k = std::min<uint64_t>(k, std::max<uint64_t>(n, 1));
  return (uint32_t)std::min<uint64_t>(k, 0xFFFFFFFFull);
}

void rh_shard_chunks(uint64_t n, uint64_t num_chunks, uint32_t n_shards, uint32_t shard, uint32_t* chunk_lo,
                     uint32_t* chunk_hi, uint64_t* row_lo, uint64_t* row_hi) {
  const uint64_t k = rh_clamp_chunks(n, num_chunks);
  const uint64_t g = std::max<uint32_t>(n_shards, 1), j = std::min<uint64_t>(shard, g - 1);
  const uint64_t c0 = k * j / g, c1 = k * (j + 1) / g;
  const uint64_t sz = n / k;
  if (chunk_lo) *chunk_lo = (uint32_t)c0;
  if (chunk_hi) *chunk_hi = (uint32_t)c1;
  if (row_lo) *row_lo = c0 * sz;
  if (row_hi) *row_hi = c1 == k ? n : c1 * sz;
}

rh_schema* rh_schema_compile(const char* json, size_t len, char** err) {
  rh_schema* out = nullptr;
  guarded(err, [&] {
    auto cs = rh::compile_schema(json, len);
    out = new rh_schema();
    out->cs = std::move(cs);
    return RH_OK;
  });
  return out;
}

void rh_schema_free(rh_schema* s) {
  if (!s) return;
  for (auto& kv : s->dev) {
    (void)hipFree(kv.second.prog);
    (void)hipFree(kv.second.sym_off);
    (void)hipFree(kv.second.sym_data);
    (void)hipFree(kv.second.desc);
    (void)hipFree(kv.second.cnt_databuf);
  }
  for (auto* table : {&s->spec, &s->espec})
    for (auto& kv : *table)
      if (kv.second)
        for (hipModule_t m : kv.second->mod)
          if (m) (void)hipModuleUnload(m);
  delete s;      // (compile jobs still running hold their own reference to `images`)
}

int rh_schema_export(const rh_schema* s, struct ArrowSchema* out) {
  if (!s || !out) return RH_ERR_ARGUMENT;
  export_field(s->cs->arrow, out);
  return RH_OK;
}

int rh_decode_device(const rh_schema* s, const void* d_data, const void* d_offsets, uint64_t data_len, uint64_t n,
                     uint64_t num_chunks, const rh_opts* opts, rh_device_result** out, rh_stats* stats, char** err) {
  if (!s || !out) return RH_ERR_ARGUMENT;
  return guarded(err, [&] {
    require_device();
    Timer t;
    // (the engine keeps bits of its own in rh_opts.flags -- RH_INTERNAL_*; a caller's stray high bits must not
    //  turn a device result into a pinned-host one or force a launch form: only the documented bits pass the C ABI)
    rh_opts pub;
    if (opts) { pub = *opts; pub.flags &= kPublicFlags; }
    *out = decode_device_impl(const_cast<rh_schema*>(s), (const uint8_t*)d_data, (const uint64_t*)d_offsets, data_len,
                              n, num_chunks, opts ? &pub : nullptr, stats);
    if (stats) stats->total_ms = t.ms();
    return RH_OK;
  });
}

char* rh_schema_kernel_source(const rh_schema* s) {
  if (!s) return nullptr;
  try {
    return dup_msg(rh::generate_kernel_source(*s->cs));
  } catch (...) {
    return nullptr;
  }
}

char* rh_schema_kernel_key(const rh_schema* s, int encode) {
  if (!s) return nullptr;
  try {
    if (encode && !s->cs->encode_unsupported.empty()) return nullptr;
    const std::string src = encode ? rh::generate_encode_source(*s->cs) : rh::generate_kernel_source(*s->cs);
    return dup_msg(rh::kernel_cache_key(src, encode != 0));
  } catch (...) {
    return nullptr;
  }
}

char* rh_schema_encode_kernel_source(const rh_schema* s) {
  if (!s || !s->cs->encode_unsupported.empty()) return nullptr;
  try {
    return dup_msg(rh::generate_encode_source(*s->cs));
  } catch (...) {
    return nullptr;
  }
}

int rh_schema_prebuild(const rh_schema* s, int* cached, char** err) {
  if (!s) return RH_ERR_ARGUMENT;
  return guarded(err, [&] {
    // every kernel of the schema, each its own compile job, side by side (kernel_jobs.h); waits for all of them
    // (RUHVRO_HIP_PREBUILD_FUSED=0: without the opt-in single-pass kernel -- the most expensive of the five, compiled on its first
    //  use otherwise; build() warms the cache that way for the schemas no single-pass test or bench line uses)
    // (RUHVRO_HIP_PREBUILD_RANGED=0: likewise without the ranged pair -- the kernels of the tiles past the LDS window, compiled when
    //  a schema first meets such tiles otherwise)
    const bool fused = env_long("RUHVRO_HIP_PREBUILD_FUSED", 1, 0, 1) != 0;
    const bool ranged = env_long("RUHVRO_HIP_PREBUILD_RANGED", 1, 0, 1) != 0;
    const unsigned parts = (rh::kDecodeParts & ~(fused ? 0u : (1u << rh::KP_FUSED)) & ~(ranged ? 0u : ((1u << rh::KP_SIZE_R) | (1u << rh::KP_EMIT_R)))) |
                           ((s->cs->encode_unsupported.empty() && !s->cs->wide) ? rh::kEncodeParts : 0u);      // (a wide schema's Arrow -> Avro pair is compiled by its first rh_encode: the encode generator unrolls every column)
    rh::KernelImage im[rh::KP_COUNT];
    const unsigned started = rh::kernel_images(s->images, *s->cs, parts, rh::CP_BLOCKING, im);
    for (int p = 0; p < rh::KP_COUNT; p++) {
      if (!(parts & (1u << p)) || im[p].state == rh::IMG_NONE) continue;
      if (im[p].state != rh::IMG_READY) throw std::runtime_error(std::string(rh::kernel_part_entry(p)) + ": " + (im[p].why.empty() ? "kernel image missing" : im[p].why));
    }
    if (cached) *cached = started == 0 ? 1 : 0;      // nothing had to be compiled
    return RH_OK;
  });
}

int rh_schema_kernels_ready(const rh_schema* s, int encode, long timeout_ms, char** err) {
  if (!s) return -1;
  try {
    const unsigned parts = encode ? rh::kEncodeParts : ((1u << rh::KP_SIZE) | (1u << rh::KP_EMIT));
    rh::KernelImage im[rh::KP_COUNT];
    rh::kernel_images(s->images, *s->cs, parts, rh::CP_CACHED_ONLY, im);      // (a first look at the disk cache; starts nothing)
    std::string why;
    const int rc = rh::kernel_images_wait(s->images, parts, timeout_ms, &why);
    if (rc < 0 && err) *err = dup_msg(why);
    return rc;
  } catch (const std::exception& e) {
    if (err) *err = dup_msg(e.what());
    return -1;
  }
}

uint32_t rh_device_result_chunks(const rh_device_result* r) { return r ? r->k : 0; }

uint64_t rh_device_result_output_bytes(const rh_device_result* r) {
  if (!r) return 0;
  try {
    settle(const_cast<rh_device_result*>(r));
    if (!r->parts.empty()) {
      uint64_t sum = 0;
      for (auto& p : r->parts) { p->tables(); sum += p->output_bytes; }
      return sum;
    }
    const_cast<rh_device_result*>(r)->tables();
  } catch (...) {
    return 0;                  // a failed asynchronous call produced nothing (rh_device_result_wait has the message)
  }
  return r->output_bytes;
}

int rh_device_result_wait(rh_device_result* r, rh_stats* stats, char** err) {
  if (!r) return RH_ERR_ARGUMENT;
  return guarded(err, [&] {
    settle(r);
    if (stats && r->has_stats) *stats = r->st;
    return RH_OK;
  });
}

int rh_device_result_export(rh_device_result* r, uint32_t chunk, struct ArrowDeviceArray* out) {
  if (!r || !out || chunk >= r->k) return RH_ERR_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  try {
    settle(r);
    rh_device_result* owner = r;
    for (size_t g = 0; g < r->parts.size(); g++)
      if (chunk >= r->part_chunk0[g] && chunk < r->part_chunk0[g] + r->parts[g]->k) { owner = r->parts[g].get(); chunk -= r->part_chunk0[g]; break; }
    owner->tables();
    export_chunk(*owner, chunk, owner->arena.ptr(), nullptr, &out->array);
  } catch (const DecodeError&) {
    return RH_ERR_DECODE;
  } catch (...) {
    return RH_ERR_RUNTIME;
  }
  out->device_id = r->device;
  out->device_type = ARROW_DEVICE_ROCM;
  out->sync_event = nullptr;   // the producing stream was synchronised before the result was settled
  return RH_OK;
}

uint32_t rh_device_result_buffers(rh_device_result* r, uint32_t chunk, uint64_t* ptrs, uint64_t* sizes, uint32_t cap) {
  if (!r || chunk >= r->k) return 0;
  try {
    settle(r);
    rh_device_result* owner = r;
    for (size_t g = 0; g < r->parts.size(); g++)
      if (chunk >= r->part_chunk0[g] && chunk < r->part_chunk0[g] + r->parts[g]->k) { owner = r->parts[g].get(); chunk -= r->part_chunk0[g]; break; }
    owner->tables();
    const uint32_t nbuf = (uint32_t)owner->cs->bufs.size();
    for (uint32_t b = 0; b < nbuf && b < cap; b++) {
      if (ptrs) ptrs[b] = (uint64_t)(uintptr_t)(owner->arena.ptr() + owner->buf_off[(size_t)b * owner->k + chunk]);
      if (sizes) sizes[b] = owner->buf_size[(size_t)b * owner->k + chunk];
    }
    return nbuf;
  } catch (...) {
    return 0;
  }
}

int rh_device_result_to_host(rh_device_result* r, struct ArrowArray* out_chunks, char** err) {
  if (!r || !out_chunks) return RH_ERR_ARGUMENT;
  return guarded(err, [&] { return to_host_impl(r, out_chunks); });
}

void rh_device_result_free(rh_device_result* r) { delete r; }

int rh_decode_packed(const rh_schema* s, const uint8_t* data, const uint64_t* offsets, uint64_t n, uint64_t num_chunks,
                     const rh_opts* opts, struct ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, char** err) {
  if (!s || !offsets || !out_chunks) return RH_ERR_ARGUMENT;
  return with_stats_line("rh_decode_packed", opts, stats, [&](const rh_opts* o, rh_stats* st) {
    return guarded(err, [&] {
      Source src;
      src.data = data;
      src.offsets = offsets;
      return decode_host_impl(const_cast<rh_schema*>(s), src, n, num_chunks, o, out_chunks, out_k, st);
    });
  });
}

int rh_decode(const rh_schema* s, const uint8_t* const* ptrs, const uint64_t* lens, uint64_t n, uint64_t num_chunks,
              const rh_opts* opts, struct ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, char** err) {
  if (!s || !out_chunks || (n && (!ptrs || !lens))) return RH_ERR_ARGUMENT;
  return with_stats_line("rh_decode", opts, stats, [&](const rh_opts* o, rh_stats* st) {
    return guarded(err, [&] {
      Source src;          // the record slices are gathered per shard, inside decode_range
      src.ptrs = ptrs;
      src.lens = lens;
      return decode_host_impl(const_cast<rh_schema*>(s), src, n, num_chunks, o, out_chunks, out_k, st);
    });
  });
}

}  // extern "C"

