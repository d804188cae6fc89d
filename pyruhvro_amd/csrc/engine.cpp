// C ABI of the direct-decode engine (include/ruhvro_hip.h): schema cache objects,
// device memory pools, the k_size -> k_scan -> k_emit launch sequence, and the
// Arrow C Data / C Device Data export of the produced buffers.
//
// Replaces, for the one hot path, the reference's chunk driver
// (ruhvro/src/deserialize.rs:76-121: pack, slice, one task per chunk, ordered
// join) -- the device boundary takes the place of the spawn_blocking boundary.
// There is NO CPU decode fallback in this library: without a HIP device every
// decode entry point fails with RH_ERR_RUNTIME.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <pthread.h>

// <hip/hip_ext.h> needs the HIP compiler; this translation unit is also built by g++ (sanitizer build).  The one
// function used from it, as declared there:
extern "C" hipError_t hipExtModuleLaunchKernel(hipFunction_t f, uint32_t globalWorkSizeX, uint32_t globalWorkSizeY,
                                               uint32_t globalWorkSizeZ, uint32_t localWorkSizeX, uint32_t localWorkSizeY,
                                               uint32_t localWorkSizeZ, size_t sharedMemBytes, hipStream_t hStream,
                                               void** kernelParams, void** extra, hipEvent_t startEvent,
                                               hipEvent_t stopEvent, uint32_t flags);

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ruhvro_hip.h"
#include "program.h"
#include "schema.h"
#include "specialize.h"
#include "kernel_jobs.h"
#include "rtc_compile.h"
#include "encode.h"

extern "C" {
// start / stop: optional hipEvent_t that receive the kernel's own begin / end timestamps (hipExtLaunchKernelGGL): no
// separate marker packets in the stream, so timing a call costs it almost nothing
int rh_launch_size(const rh::KParams* P, uint32_t lds_bytes, void* stream, void* start, void* stop);
int rh_launch_scan_layout(const rh::KParams* P, const rh::LParams* L, void* stream, void* start, void* stop);
int rh_launch_scan(const rh::KParams* P, void* stream, void* start, void* stop);
int rh_launch_init(void* const* bufptr, const uint64_t* bufsize, const rh::BufDesc* desc, uint32_t nbuf, uint32_t k,
                   const unsigned long long* ctrl, void* stream);
int rh_launch_layout(const rh::LParams* L, void* stream);
int rh_launch_publish(void* ctrl, void* host, uint32_t head_words, uint32_t null_entries, uint32_t flag_word, uint32_t token, uint32_t nslots, void* stream);
int rh_launch_emit(const rh::KParams* P, uint32_t lds_bytes, void* stream, void* start, void* stop);
int rh_set_max_lds(uint32_t bytes);
uint32_t rh_lds_fixed_bytes(int K, int list_depth, int nnodes, int nbuf);
// Arrow -> Avro kernels (encode.hip)
int rh_launch_esize(const rh::EParams* P, uint32_t lds_bytes, void* stream);
int rh_launch_eemit(const rh::EParams* P, uint32_t lds_bytes, void* stream);
uint32_t rh_enc_lds_bytes(int ndom, int list_depth);
}
#include <memory>

namespace {

using rh::CompiledSchema;
using rh::DecNode;

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct DecodeError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct ValueClassError : std::runtime_error {   // data-dependent failures of other paths (encode): ValueError in Python
  using std::runtime_error::runtime_error;
};
struct NeedWideIndex {};   // a chunk buffer reaches 4 GiB: only the generic kernels index that far
struct NeedTwoPass {};     // the single-pass form outgrew a column capacity (or needs what only the two-pass layout checks): repeat

std::atomic<uint64_t> g_counters[RH_CTR_COUNT];     // rh_engine_counters (include/ruhvro_hip.h)
inline void count(int which) { g_counters[which].fetch_add(1, std::memory_order_relaxed); }

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw HipError(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);      \
  } while (0)

char* dup_msg(const std::string& s) {
  char* p = (char*)std::malloc(s.size() + 1);
  if (p) std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

constexpr uint64_t kAlign = 256;
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// caching device / pinned-host pools (one per process; blocks are reused across calls so a
// steady-state decode does no hipMalloc)
// ---------------------------------------------------------------------------
struct Block {
  void* p = nullptr;
  uint64_t size = 0;
  int device = 0;
};

class Pool {
 public:
  explicit Pool(bool host) : host_(host) {
    // cached (idle) bytes this pool may hold on to between calls.  Pinned host memory is the scarcer resource: a
    // long-lived process should not keep tens of GiB page-locked because of one large call.
    const char* e = std::getenv(host ? "RUHVRO_HIP_PINNED_CACHE_MB" : "RUHVRO_HIP_DEVICE_CACHE_MB");
    max_cached_ = e ? ((uint64_t)std::strtoull(e, nullptr, 10) << 20) : (host ? (4ull << 30) : (24ull << 30));
  }
  Block get(uint64_t size, int device) {
    size = align_up(std::max<uint64_t>(size, 1), 1 << 16);
    {
      std::lock_guard<std::mutex> g(mu_);
      int best = -1;
      for (size_t i = 0; i < free_.size(); i++) {
        if ((host_ || free_[i].device == device) && free_[i].size >= size && free_[i].size <= size * 2 + (1 << 20)) {
          if (best < 0 || free_[i].size < free_[best].size) best = (int)i;
        }
      }
      if (best >= 0) {
        Block b = free_[best];
        free_.erase(free_.begin() + best);
        cached_ -= b.size;
        poison(b);
        return b;
      }
    }
    Block b;
    b.size = size;
    b.device = device;
    hipError_t e = host_ ? hipHostMalloc(&b.p, size, hipHostMallocDefault) : hipMalloc(&b.p, size);
    if (e != hipSuccess) {
      trim(0);
      e = host_ ? hipHostMalloc(&b.p, size, hipHostMallocDefault) : hipMalloc(&b.p, size);
    }
    if (e != hipSuccess) throw HipError(std::string("HIP allocation of ") + std::to_string(size) + " bytes failed: " + hipGetErrorString(e));
    poison(b);
    return b;
  }
  // RUHVRO_HIP_POISON=1 (test mode): every block is handed out filled with 0xA5, so that nothing can lean on what a block
  // happens to hold -- fresh allocations read as zero, which hides a read of padding or of a slot nobody wrote until the
  // pool hands out used memory (the encode kernels' look-ahead found that way: profiles/r04zg_*).  Slow: a synchronous fill.
  void poison(const Block& b) const {
    static const bool on = [] { const char* e = std::getenv("RUHVRO_HIP_POISON"); return e && *e && *e != '0'; }();
    if (!on || !b.p) return;
    if (host_) { std::memset(b.p, 0xA5, b.size); return; }
    (void)hipMemset(b.p, 0xA5, b.size);
    (void)hipDeviceSynchronize();
  }
  // A cached block of a suitable size, or an empty Block: never allocates.
  Block try_get(uint64_t size, int device) {
    size = align_up(std::max<uint64_t>(size, 1), 1 << 16);
    std::lock_guard<std::mutex> g(mu_);
    int best = -1;
    for (size_t i = 0; i < free_.size(); i++)
      if ((host_ || free_[i].device == device) && free_[i].size >= size && free_[i].size <= size * 2 + (1 << 20))
        if (best < 0 || free_[i].size < free_[best].size) best = (int)i;
    if (best < 0) return Block();
    Block b = free_[best];
    free_.erase(free_.begin() + best);
    cached_ -= b.size;
    poison(b);
    return b;
  }
  void put(Block b) {
    if (!b.p) return;
    {
      std::lock_guard<std::mutex> g(mu_);
      free_.push_back(b);
      cached_ += b.size;
    }
    trim(max_cached_);
  }
  void trim(uint64_t keep) {
    std::vector<Block> drop;
    {
      std::lock_guard<std::mutex> g(mu_);
      while (cached_ > keep && !free_.empty()) {
        size_t big = 0;
        for (size_t i = 1; i < free_.size(); i++)
          if (free_[i].size > free_[big].size) big = i;
        drop.push_back(free_[big]);
        cached_ -= free_[big].size;
        free_.erase(free_.begin() + big);
      }
    }
    for (auto& b : drop) {
      if (host_) (void)hipHostFree(b.p);
      else (void)hipFree(b.p);
    }
  }

 private:
  uint64_t max_cached_;
  bool host_;
  std::mutex mu_;
  std::vector<Block> free_;
  uint64_t cached_ = 0;
};

Pool& dev_pool() { static Pool* p = new Pool(false); return *p; }
Pool& pin_pool() { static Pool* p = new Pool(true); return *p; }

struct Lease {   // RAII pool block
  Pool* pool = nullptr;
  Block b;
  Lease() = default;
  Lease(Pool& p, uint64_t size, int device) : pool(&p), b(p.get(size, device)) {}
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;
  Lease(Lease&& o) noexcept : pool(o.pool), b(o.b) { o.pool = nullptr; o.b = Block(); }
  Lease& operator=(Lease&& o) noexcept {
    if (this != &o) { release(); pool = o.pool; b = o.b; o.pool = nullptr; o.b = Block(); }
    return *this;
  }
  ~Lease() { release(); }
  void release() { if (pool && b.p) pool->put(b); pool = nullptr; b = Block(); }
  uint8_t* ptr() const { return (uint8_t*)b.p; }
};

// Control blocks of the decode calls (first_bad, layout flag, ticket, null counts, chunk totals -- program.h): handed
// out ALL ZERO and zeroed again when they come back -- asynchronously, on the stream of the call that used them.  The
// next call on that stream is ordered behind that memset, so no call has a memset in front of its first kernel any
// more (2 us of fill kernel + the gap behind it, at the head of every call: profiles/r03e_timeline_*.txt).
class CtrlPool {
 public:
  // clean: the call's last kernel (rh_k_publish) already left the block zeroed -- no memset on the way back
  struct Blk { void* p = nullptr; uint64_t size = 0; int device = 0; hipStream_t stream = nullptr; bool clean = false; };
  Blk get(uint64_t size, int device, hipStream_t stream) {
    size = align_up(std::max<uint64_t>(size, 1), 4096);
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < free_.size(); i++)
        if (free_[i].device == device && free_[i].stream == stream && free_[i].size == size) {
          Blk b = free_[i];
          free_.erase(free_.begin() + (long)i);
          return b;
        }
    }
    Blk b;
    b.size = size; b.device = device; b.stream = stream;
    hipError_t e = hipMalloc(&b.p, size);
    if (e != hipSuccess) throw HipError(std::string("HIP allocation of a control block failed: ") + hipGetErrorString(e));
    e = hipMemsetAsync(b.p, 0, size, stream);          // ordered before the kernels of the call that asked for it
    if (e != hipSuccess) { (void)hipFree(b.p); throw HipError(std::string("hipMemsetAsync failed: ") + hipGetErrorString(e)); }
    return b;
  }
  void put(Blk b) {
    if (!b.p) return;
    if (!b.clean && hipMemsetAsync(b.p, 0, b.size, b.stream) != hipSuccess) { (void)hipFree(b.p); return; }
    b.clean = false;
    Blk drop;
    {
      std::lock_guard<std::mutex> g(mu_);
      free_.push_back(b);
      if (free_.size() > 64) { drop = free_.front(); free_.erase(free_.begin()); }   // streams that went away
    }
    if (drop.p) (void)hipFree(drop.p);
  }

 private:
  std::mutex mu_;
  std::vector<Blk> free_;
};
CtrlPool& ctrl_pool() { static CtrlPool* p = new CtrlPool(); return *p; }

struct CtrlLease {
  CtrlPool::Blk b;
  CtrlLease(uint64_t size, int device, hipStream_t stream) : b(ctrl_pool().get(size, device, stream)) {}
  CtrlLease(const CtrlLease&) = delete;
  CtrlLease& operator=(const CtrlLease&) = delete;
  ~CtrlLease() { ctrl_pool().put(b); }
  uint8_t* ptr() const { return (uint8_t*)b.p; }
};

// ---------------------------------------------------------------------------
// compiled schema + its per-device copy
// ---------------------------------------------------------------------------
struct DeviceProgram {
  rh::Op* prog = nullptr;
  uint32_t* sym_off = nullptr;
  uint8_t* sym_data = nullptr;
  rh::BufDesc* desc = nullptr;
  int32_t* cnt_databuf = nullptr;
};

struct SpecKernel {      // schema-specialised kernels loaded on one device (each kernel is its own code object, kernel_jobs.h)
  hipModule_t mod[3] = {nullptr, nullptr, nullptr};
  hipFunction_t size_fn = nullptr, emit_fn = nullptr;
  // the single-pass form (decode kernels only): compiled and loaded when a call first asks for it, so it is written while
  // other calls of the schema read it
  std::atomic<hipFunction_t> fused_fn{nullptr};
  bool fused_dead = false;  // no such kernel for this schema (K > 64), or its compile failed
  bool ok = false;          // size_fn and emit_fn are loaded
  bool dead = false;        // they never will be: `why` says why (a failure is remembered)
  std::string why;
};

}  // namespace

struct rh_schema {
  std::unique_ptr<CompiledSchema> cs;
  std::mutex mu;
  std::map<int, DeviceProgram> dev;
  std::map<int, std::unique_ptr<SpecKernel>> spec;
  std::map<int, std::unique_ptr<SpecKernel>> espec;   // Arrow -> Avro kernels (rh_espec_size / rh_espec_emit)
  std::shared_ptr<rh::KernelImages> images = rh::new_kernel_images();   // their code objects (device independent) + compile jobs
  // Arena bytes per (payload byte + 64 B per record) that the last decode of this schema needed: sizes the arena of
  // the next call BEFORE its totals are known, so that the call is one stream submission (decode_device_impl1).
  // 0 = no history yet (the first call of a schema lays its arena out on the host, after the scan).
  std::atomic<double> arena_ratio{0.0};
  // Single-pass form: what every counter's column needed PER ROW of a chunk in the last settled call (the largest chunk's
  // figure): sizes each column's capacity before the launch (rh_decode_call::try_single).  Empty = no history.  A call that
  // outgrows its capacities is repeated on the two-pass form and the schema sits the next calls out (backing off: data that
  // keeps changing character stays on the two-pass form, one outlier batch costs eight calls).
  std::vector<double> per_row;
  uint32_t single_cooldown = 0, single_backoff = 0;      // calls the single pass sits out after a fail-over (8, 16, ... 1024; a success clears it)
};

namespace {

const DeviceProgram& device_program(rh_schema* s, int device) {
  std::lock_guard<std::mutex> g(s->mu);
  auto it = s->dev.find(device);
  if (it != s->dev.end()) return it->second;
  const CompiledSchema& cs = *s->cs;
  DeviceProgram d;
  const size_t pb = cs.prog.size() * sizeof(rh::Op), so = cs.sym_off.size() * 4, sd = cs.sym_data.size(),
               bd = std::max<size_t>(cs.bufs.size(), 1) * sizeof(rh::BufDesc);
  HIPCHK(hipMalloc((void**)&d.prog, pb));
  HIPCHK(hipMalloc((void**)&d.sym_off, so));
  HIPCHK(hipMalloc((void**)&d.sym_data, sd));
  HIPCHK(hipMalloc((void**)&d.desc, bd));
  HIPCHK(hipMemcpy(d.prog, cs.prog.data(), pb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.sym_off, cs.sym_off.data(), so, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.sym_data, cs.sym_data.data(), sd, hipMemcpyHostToDevice));
  if (!cs.bufs.empty()) HIPCHK(hipMemcpy(d.desc, cs.bufs.data(), cs.bufs.size() * sizeof(rh::BufDesc), hipMemcpyHostToDevice));
  std::vector<int32_t> cdb((size_t)std::max(cs.K, 1), -1);
  for (size_t b = 0; b < cs.bufs.size(); b++)
    if (cs.bufs[b].kind == rh::BK_DATA) cdb[cs.bufs[b].counter] = (int32_t)b;
  HIPCHK(hipMalloc((void**)&d.cnt_databuf, cdb.size() * 4));
  HIPCHK(hipMemcpy(d.cnt_databuf, cdb.data(), cdb.size() * 4, hipMemcpyHostToDevice));
  return s->dev.emplace(device, d).first->second;
}

uint64_t spec_min_records() {
  static const uint64_t v = [] {
    const char* e = std::getenv("RUHVRO_HIP_SPECIALIZE_MIN");
    return e ? (uint64_t)std::strtoull(e, nullptr, 10) : (uint64_t)32768;
  }();
  return v;
}

hipFunction_t load_part(SpecKernel& k, int slot, const rh::KernelImage& im, int part) {
  hipError_t e = hipModuleLoadData(&k.mod[slot], im.code->data());
  if (e != hipSuccess) throw std::runtime_error(std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  hipFunction_t fn = nullptr;
  e = hipModuleGetFunction(&fn, k.mod[slot], rh::kernel_part_entry(part));
  if (e != hipSuccess) throw std::runtime_error(std::string("hipModuleGetFunction: ") + hipGetErrorString(e));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipGetLastError();
  return fn;
}

// Specialised kernels of this schema on `device`.  Their code objects come from the kernel cache; on a miss the policy
// decides: nothing (small calls), compile jobs in the background -- THIS call then runs on the generic kernels and a later
// one finds the objects ready -- or wait for the jobs (RH_KERNEL_SPECIALIZED, an explicit request).  The schema's mutex is
// only ever held for table look-ups and hipModuleLoadData, never across a compile.  `want_fused`: also the single-pass
// kernel (compiled on first request).  A failure is remembered (dead, why).
const SpecKernel& spec_kernel(rh_schema* s, int device, rh::CompilePolicy policy, bool encode = false, bool want_fused = false) {
  std::map<int, std::unique_ptr<SpecKernel>>& table = encode ? s->espec : s->spec;
  SpecKernel* k = nullptr;
  {
    std::lock_guard<std::mutex> g(s->mu);
    std::unique_ptr<SpecKernel>& slot = table[device];
    if (!slot) slot.reset(new SpecKernel);
    k = slot.get();                                   // (entries are never removed while the schema lives)
    if (k->dead) return *k;
    if (k->ok && (!want_fused || k->fused_dead || k->fused_fn.load(std::memory_order_acquire))) return *k;
  }
  const int p_size = encode ? rh::KP_ESIZE : rh::KP_SIZE, p_emit = encode ? rh::KP_EEMIT : rh::KP_EMIT;
  const unsigned parts = (1u << p_size) | (1u << p_emit) | ((want_fused && !encode) ? (1u << rh::KP_FUSED) : 0u);
  rh::KernelImage im[rh::KP_COUNT];
  const unsigned started = rh::kernel_images(s->images, *s->cs, parts, policy, im);     // (blocks only under CP_BLOCKING)
  if (started && policy == rh::CP_BACKGROUND) g_counters[RH_CTR_BACKGROUND_COMPILES].fetch_add(started, std::memory_order_relaxed);
  std::lock_guard<std::mutex> g(s->mu);
  if (k->dead) return *k;
  try {
    if (!k->ok) {
      for (int p : {p_size, p_emit})
        if (im[p].state == rh::IMG_FAILED || im[p].state == rh::IMG_NONE) {
          k->dead = true;
          k->why = im[p].state == rh::IMG_NONE ? std::string("the schema has no such kernel") : im[p].why;
          return *k;
        }
      if (im[p_size].state == rh::IMG_READY && im[p_emit].state == rh::IMG_READY) {
        k->size_fn = load_part(*k, 0, im[p_size], p_size);
        k->emit_fn = load_part(*k, 1, im[p_emit], p_emit);
        k->ok = true;
        k->why.clear();
      } else {
        k->why = (im[p_size].state == rh::IMG_COMPILING || im[p_emit].state == rh::IMG_COMPILING) ? "compiling" : "not cached";
      }
    }
    if (k->ok && want_fused && !encode && !k->fused_dead && !k->fused_fn.load(std::memory_order_relaxed)) {
      const rh::KernelImage& f = im[rh::KP_FUSED];
      if (f.state == rh::IMG_FAILED || f.state == rh::IMG_NONE) k->fused_dead = true;
      else if (f.state == rh::IMG_READY) k->fused_fn.store(load_part(*k, 2, f, rh::KP_FUSED), std::memory_order_release);
    }
  } catch (const std::exception& e) {
    k->dead = true;
    k->ok = false;
    k->why = e.what();
  }
  return *k;
}

// What a call of `n` records may spend on kernels this schema does not have yet.
rh::CompilePolicy compile_policy(int mode, uint64_t n) {
  if (mode == RH_KERNEL_SPECIALIZED) return rh::CP_BLOCKING;
  // RUHVRO_HIP_SYNC_COMPILE=1: the pre-ABI-6 behaviour -- a large call waits for its schema's compile (deterministic benchmarks
  // of a cold process, nothing else)
  static const bool sync = [] { const char* e = std::getenv("RUHVRO_HIP_SYNC_COMPILE"); return e && *e && *e != '0'; }();
  if (n >= spec_min_records()) return sync ? rh::CP_BLOCKING : rh::CP_BACKGROUND;
  return rh::CP_CACHED_ONLY;
}

int launch_module(hipFunction_t f, const rh::KParams& P, uint32_t grid, uint32_t block, uint32_t lds, hipStream_t stream,
                  hipEvent_t start = nullptr, hipEvent_t stop = nullptr) {
  rh::KParams copy = P;
  void* args[] = {&copy};
  if (!start && !stop) return (int)hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, stream, args, nullptr);
  // the kernel's own start / stop timestamps land in the events (global size is in work-items here)
  return (int)hipExtModuleLaunchKernel(f, grid * block, 1, 1, block, 1, 1, lds, stream, args, nullptr, start, stop, 0);
}

std::string format_error(const rh::ErrInfo& e) {
  char buf[128];
  switch (e.code) {
    case rh::E_EOB: return "unexpected end of buffer";
    case rh::E_VARINT: return "zigzag varint too long";
    case rh::E_EOB_F32: return "unexpected end of buffer (f32)";
    case rh::E_EOB_F64: return "unexpected end of buffer (f64)";
    case rh::E_BOOL: std::snprintf(buf, sizeof buf, "invalid boolean byte: %lld", (long long)e.detail); return buf;
    case rh::E_NEGLEN: return "negative string length";
    case rh::E_EOB_STR: return "unexpected end of buffer (string)";
    case rh::E_ENUM: std::snprintf(buf, sizeof buf, "enum index %llu out of range", (unsigned long long)e.detail); return buf;
    case rh::E_BRANCH: std::snprintf(buf, sizeof buf, "invalid union branch index: %lld", (long long)e.detail); return buf;
    case rh::E_UNION: std::snprintf(buf, sizeof buf, "union branch index out of range: %lld", (long long)e.detail); return buf;
    case rh::E_LIST_RANGE:
      std::snprintf(buf, sizeof buf, "array/map block count %lld of zero-width items exceeds the supported range", (long long)e.detail);
      return buf;
    case rh::E_INTERNAL: return "internal error: the fast and the careful walk disagree on a record";
    case rh::E_EOB_FIXED: return "unexpected end of buffer (fixed)";
    case rh::E_DECIMAL: std::snprintf(buf, sizeof buf, "decimal value of %lld bytes does not fit Decimal128", (long long)e.detail); return buf;
    case rh::E_UUID: return "invalid uuid string";
    case rh::E_DURATION: std::snprintf(buf, sizeof buf, "duration with %lld months has no value in Duration(ms)", (long long)e.detail); return buf;
    default: return "decode error";
  }
}

// roctx ranges around the stages of a call (gather, H2D, kernels, D2H, export) so that a rocprofv3 --marker-trace
// timeline shows them.  The marker library is bound at run time: the one the profiler already loaded (RTLD_NOLOAD),
// or, with RUHVRO_HIP_ROCTX=1, loaded by name; without either the ranges cost one predictable branch.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = std::getenv("RUHVRO_HIP_ROCTX");
    const bool want = e && *e && *e != '0';
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_LAZY | RTLD_NOLOAD);
      if (!h && want) h = dlopen(name, RTLD_LAZY);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
  static const Roctx& get() { static const Roctx r; return r; }
};
struct Range {
  bool on;
  explicit Range(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
  ~Range() { if (on) Roctx::get().pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};

// Host-side phase times of one call (RUHVRO_HIP_HOSTPROF=1 -> one stderr line per decode_device call): where the
// microseconds between the kernels go on small inputs.
struct HostProf {
  bool on;
  std::chrono::steady_clock::time_point t0;
  std::string line;
  HostProf() {
    static const bool e = [] { const char* v = std::getenv("RUHVRO_HIP_HOSTPROF"); return v && *v && *v != '0'; }();
    on = e;
    if (on) t0 = std::chrono::steady_clock::now();
  }
  void mark(const char* what) {
    if (!on) return;
    const auto t = std::chrono::steady_clock::now();
    char buf[64];
    std::snprintf(buf, sizeof buf, " %s=%.1f", what, std::chrono::duration<double, std::micro>(t - t0).count());
    line += buf;
    t0 = t;
  }
  ~HostProf() { if (on) std::fprintf(stderr, "[ruhvro_hip hostprof us]%s\n", line.c_str()); }
};

// RUHVRO_HIP_TIMELINE=1: one stderr line per stage boundary of every shard of a host call, in ms since the call began
// (when the gather of a shard ended, when it held each PCIe direction): shows where a pipelined call waits.
struct Timeline {
  static bool on() {
    static const bool e = [] { const char* v = std::getenv("RUHVRO_HIP_TIMELINE"); return v && *v && *v != '0'; }();
    return e;
  }
  static std::chrono::steady_clock::time_point& t0() {
    static std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    return t;
  }
  static void start() { if (on()) t0() = std::chrono::steady_clock::now(); }
  static void mark(uint32_t shard, const char* what) {
    if (!on()) return;
    std::fprintf(stderr, "[ruhvro_hip timeline] %8.3f ms  shard %u  %s\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0()).count(), shard, what);
  }
};

rh_opts default_opts() {
  rh_opts o;
  std::memset(&o, 0, sizeof o);
  o.device = -1;
  return o;
}

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  float ms() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace

// ---------------------------------------------------------------------------
// device result
// ---------------------------------------------------------------------------
struct rh_decode_call;                 // one device-resident decode call (DeviceDecode below), still on its stream
struct rh_device_result {
  const CompiledSchema* cs = nullptr;
  int device = 0;
  uint64_t n = 0, sz = 0, rows_last = 0;
  uint32_t k = 1;
  Lease arena;                         // all Arrow buffers of all chunks
  uint64_t arena_bytes = 0;
  std::vector<uint64_t> buf_off;       // [nbuf][k] offset into arena
  std::vector<uint64_t> buf_size;      // [nbuf][k] allocated bytes
  std::vector<uint64_t> dom_rows;      // [ndom][k]
  std::vector<uint64_t> data_bytes;    // [K][k] totals
  std::vector<uint64_t> layout_bytes;  // [K][k] single-pass form: the CAPACITIES the arena was laid out with (empty: laid out exactly)
  std::vector<uint32_t> nullcount;     // [nnodes][k]
  uint64_t output_bytes = 0;           // exact (unpadded) Arrow bytes
  // The [buf][chunk] tables above are a pure function of (schema, chunk geometry, data_bytes).  A call whose arena was
  // laid out on the device (and accepted) leaves them to the first reader: tables() -- export, host copy, byte counts.
  std::mutex tables_mu;
  bool tables_done = false;
  // RH_ASYNC: the call is on its stream but the host has not looked at its outcome yet (settle(), below DeviceDecode)
  std::unique_ptr<rh_decode_call> pending;
  std::exception_ptr fail;             // what settle() found: every later accessor reports it again
  rh_stats st;                         // stage timings of an asynchronous call that asked for them
  bool has_stats = false;
  // A call that dealt its chunk groups to internal streams (decode_device_split): one complete result per group, in
  // chunk order; part g holds chunks [part_chunk0[g], part_chunk0[g + 1]) of this call.  The fields above other than
  // cs / device / n / k / sz / rows_last / fail are then unused.
  std::vector<std::unique_ptr<rh_device_result>> parts;
  std::vector<uint32_t> part_chunk0;
  std::vector<hipEvent_t> join_events;  // recorded on the internal streams, waited for by the caller's stream (recycled on free)

  rh_device_result();
  ~rh_device_result();
  uint64_t rows(int dom, uint32_t c) const { return dom_rows[(size_t)dom * k + c]; }
  void fill_tables();                  // host statement of the layout rule (program.h buf_bytes / buf_slot_bytes)
  void tables() { std::lock_guard<std::mutex> g(tables_mu); if (!tables_done) fill_tables(); }
};

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

struct rh_device_encoded {             // result of rh_encode_device: k BinaryArrays in HBM
  int device = 0;
  uint64_t n = 0, sz = 0, rows_last = 0;
  uint32_t k = 1;
  Lease out;                           // per chunk: i32 offsets[rows + 1] | data
  uint64_t out_bytes = 0;              // bytes of `out` in use
  std::vector<uint64_t> ooff;          // [k][2] offsets of the two buffers
  std::vector<uint64_t> data_bytes;    // [k] Avro bytes per chunk
  uint64_t exact = 0;
  uint64_t rows(uint32_t c) const { return n == 0 ? 0 : (c == k - 1 ? rows_last : sz); }
};

namespace {

// ---------------------------------------------------------------------------
// Arrow C Data export
// ---------------------------------------------------------------------------
struct Slab {   // host copy of the arena, shared by the k chunk arrays (freed when the last one is released)
  std::atomic<int> refs{0};
  void* base = nullptr;
  Block pinned;           // large results live in pooled pinned memory: the D2H copy runs at PCIe speed
  void free_mem() {
    if (pinned.p) {
      pinned_result_bytes().fetch_sub(pinned.size);
      pin_pool().put(pinned);
    } else {
      std::free(base);
    }
    pinned = Block();
    base = nullptr;
  }
  static std::atomic<uint64_t>& pinned_result_bytes() {   // pinned memory currently lent to live results
    static std::atomic<uint64_t> v{0};
    return v;
  }
};

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

// ---------------------------------------------------------------------------
// the launch sequence
// ---------------------------------------------------------------------------
// integer knob from the environment, read at every use (tests change them inside one process); out of range = default
constexpr long kSinglePassDefault = 0;           // RUHVRO_HIP_SINGLE_PASS: 1 = every qualifying call prefers the single-pass form (else RH_SINGLE_PASS per call)
constexpr int RH_INTERNAL_TWO_PASS = 0x100;      // rh_opts.flags, engine-internal: this call must take the two-pass path
constexpr long kInternalStreamsDefault = 1;      // RUHVRO_HIP_INTERNAL_STREAMS (decode_device_split)
constexpr long kSplitMinDefault = 1000000;       // RUHVRO_HIP_SPLIT_MIN: records below which a call is never split

long env_long(const char* name, long dflt, long lo, long hi) {
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = std::strtol(e, &end, 10);
  return (end && *end == 0 && v >= lo && v <= hi) ? v : dflt;
}

struct Events {
  hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool on = false;
  int device = 0;
  // events are recycled per device: creating and destroying six of them was a third of what a timed call cost
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::map<int, std::vector<hipEvent_t>>& idle() { static auto* v = new std::map<int, std::vector<hipEvent_t>>(); return *v; }
  void init() {
    HIPCHK(hipGetDevice(&device));
    {
      std::lock_guard<std::mutex> g(mu());
      auto& v = idle()[device];
      for (auto& x : e)
        if (!v.empty()) { x = v.back(); v.pop_back(); }
    }
    for (auto& x : e)
      if (!x) HIPCHK(hipEventCreate(&x));
    on = true;
  }
  ~Events() {
    if (!on) return;
    std::lock_guard<std::mutex> g(mu());
    auto& v = idle()[device];
    for (auto& x : e) {
      if (!x) continue;
      if (v.size() < 64) v.push_back(x);
      else (void)hipEventDestroy(x);
    }
  }
  void rec(int i, hipStream_t s) { if (on) HIPCHK(hipEventRecord(e[i], s)); }
  hipEvent_t at(int i) const { return on ? e[i] : nullptr; }
  float ms(int a, int b) {
    float t = 0;
    if (on && hipEventElapsedTime(&t, e[a], e[b]) != hipSuccess) { t = 0; (void)hipGetLastError(); }   // (a pair that was never recorded: no sticky error left behind)
    return t;
  }
};

// "This call's work is done" markers of asynchronous calls (RH_ASYNC): hipStreamSynchronize would also wait for every
// LATER call on the stream -- exactly the calls the asynchronous form exists to keep queued.  Recycled per device.
struct DoneEvent {
  hipEvent_t e = nullptr;
  int device = 0;
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::map<int, std::vector<hipEvent_t>>& idle() { static auto* v = new std::map<int, std::vector<hipEvent_t>>(); return *v; }
  void record(int dev, hipStream_t s) {
    device = dev;
    {
      std::lock_guard<std::mutex> g(mu());
      auto& v = idle()[device];
      if (!v.empty()) { e = v.back(); v.pop_back(); }
    }
    if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, s));
  }
  void wait() { if (e) HIPCHK(hipEventSynchronize(e)); }
  ~DoneEvent() {
    if (!e) return;
    std::lock_guard<std::mutex> g(mu());
    auto& v = idle()[device];
    if (v.size() < 64) v.push_back(e);
    else (void)hipEventDestroy(e);
  }
};

// Chunk geometry of a call that decodes a contiguous RANGE of another call's chunks (the pipelined host path):
// k chunks of sz rows, the last one rows_last, instead of the split derived from (n, num_chunks).
struct ChunkGeo {
  uint64_t sz, rows_last;
  uint32_t k;
  uint64_t payload_bytes;   // bytes of the range's records (data_len is the absolute end offset there)
};

rh_device_result* decode_device_impl1(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len,
                                      uint64_t n, uint64_t num_chunks, const rh_opts* opts, rh_stats* stats, const ChunkGeo* geo);

rh_device_result* decode_device_split(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len, uint64_t n,
                                      const rh_opts& opts, uint32_t k, uint64_t sz, uint64_t rows_last, unsigned G);

rh_device_result* decode_device_impl(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len,
                                     uint64_t n, uint64_t num_chunks, const rh_opts* opts, rh_stats* stats,
                                     const ChunkGeo* geo = nullptr) {
  try {
  try {
    // in-call overlap (decode_device_split): a large call deals its chunk groups to internal streams
    const long G = env_long("RUHVRO_HIP_INTERNAL_STREAMS", kInternalStreamsDefault, 1, 8);
    const long groups_env = env_long("RUHVRO_HIP_SPLIT_GROUPS", 0, 0, 64);      // (> 1 with one stream: the groups run back to back)
    if ((G > 1 || groups_env > 1) && !geo && !stats && n >= (uint64_t)env_long("RUHVRO_HIP_SPLIT_MIN", kSplitMinDefault, 1, 1l << 40) &&
        (!opts || (opts->flags & 3) != RH_KERNEL_GENERIC) && s->arena_ratio.load() > 0 && env_long("RUHVRO_HIP_TWO_SYNC", 0, 0, 1) == 0 &&
        env_long("RUHVRO_HIP_ARENA_PERMILLE", -1, 0, 1000000) < 0) {
      const rh_opts o = opts ? *opts : default_opts();
      uint64_t k64 = rh_clamp_chunks(n, num_chunks), sz = n / std::max<uint64_t>(k64, 1);
      bool ok = true;
      if (o.chunk_rows) {        // explicit geometry: validated by the unsplit path when it does not hold
        ok = num_chunks >= 1 && num_chunks <= 0xFFFFFFFFull && (num_chunks - 1) <= n / o.chunk_rows &&
             !(n > 0 && n == (num_chunks - 1) * o.chunk_rows && num_chunks > 1);
        k64 = num_chunks; sz = o.chunk_rows;
      }
      const unsigned g = (unsigned)std::min<uint64_t>((uint64_t)G, k64);
      if (ok && (g > 1 || (groups_env > 1 && k64 > 1)) && !((uintptr_t)d_data & 15)) {
        const uint64_t rows_last = n - (k64 - 1) * sz;
        try {
          if (rh_device_result* r = decode_device_split(s, d_data, d_offsets, data_len, n, o, (uint32_t)k64, sz, rows_last, g)) return r;
        } catch (const NeedWideIndex&) {
          // (a child row domain beyond 32-bit indexing: the whole call goes to the generic kernels below, unsplit)
          count(RH_CTR_WIDE_FALLBACKS);
          rh_opts o2 = o;
          o2.flags = (o.flags & ~3) | RH_KERNEL_GENERIC;
          return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o2, stats, geo);
        }
      }
    }
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, opts, stats, geo);
  } catch (const NeedTwoPass&) {
    rh_opts o = default_opts();
    if (opts) o = *opts;
    o.flags |= RH_INTERNAL_TWO_PASS;
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o, stats, geo);
  }
  } catch (const NeedWideIndex&) {
    count(RH_CTR_WIDE_FALLBACKS);
    rh_opts o = default_opts();
    if (opts) o = *opts;
    o.flags = RH_KERNEL_GENERIC;
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o, stats, geo);
  }
}

}  // namespace

// One device-resident decode call.  enqueue() puts the whole call on the stream (k_size -> k_scan+k_layout -> k_init ->
// k_emit -> one D2H of the control words) and finish() waits for it and settles the result (error check, arena
// retry, host tables).  rh_decode_device runs both back to back; with RH_ASYNC the result is handed out between the
// two and rh_device_result_wait() (or the first accessor that needs a host-side fact) runs finish(): the caller's next
// call is on the stream before this one has drained, which is what a pipeline of small batches needs -- a 1M-record
// call is 0.15 ms of kernels behind ~25 us of host turn-around (profiles/r03q_timeline_*.txt).
struct rh_decode_call {
  // the call
  rh_schema* s;
  const CompiledSchema& cs;
  const uint8_t* d_data;
  const uint64_t* d_offsets;
  uint64_t data_len, n, num_chunks;
  rh_opts opts;                 // by value: an asynchronous call outlives the caller's struct
  bool want_stats;
  rh_stats st;
  rh_device_result& r;
  HostProf hp;
  int device = 0;
  hipStream_t stream = nullptr;
  ChunkGeo geo_v;
  const ChunkGeo* geo = nullptr;
  // derived
  uint32_t k = 1;
  int K = 0, nnodes = 0, nbuf = 0;
  const DeviceProgram* dp = nullptr;
  const SpecKernel* sk = nullptr;
  uint64_t narrow_rows = 0, tile = 0, bpc64 = 0, payload = 0;
  uint32_t nblocks = 0;
  uint64_t o_null = 0, o_tot = 32, ctrl_bytes = 0;
  uint32_t null_slots = rh::kNullSlots;      // program.h null_slots_for(k)
  Lease ws, hctrl, dtab, prof_buf;
  std::unique_ptr<CtrlLease> ctrl;
  rh::KParams P;
  uint32_t lds_bytes = 0, emit_lds = 0;
  bool profile = false;
  Events ev;
  std::vector<uint64_t> totals;
  uint64_t n_entries = 0, tab_bytes = 0;
  uint64_t* d_sizes = nullptr;
  uint64_t exact = 0;
  bool child_bitmaps = false, fused = false, timed_size = false;
  bool range_of_host_call = false;   // a chunk range of a host call (decode_range) or a group of a split call: the caller gave the geometry
  bool single = false;          // the single-pass form ran (rh_spec_fused): arena laid out from capacities
  std::vector<uint64_t> caps;   // [K][k] those capacities
  uint64_t arena_cap = 0, o_tick = 0;
  Lease lookback, hcaps;
  double basis = 0;
  bool settled = false;         // finish() ran (or the call completed inside enqueue())
  bool async = false;           // RH_ASYNC: the call is settled later; without rh_k_publish its end is marked with a DoneEvent
  DoneEvent done;
  // in-call overlap (decode_device_split, staggered form): this group's size pass starts behind `start_after` (the previous
  // group's size pass) and marks its own end with `sized`, so that size pass g+1 runs beside emit pass g
  hipEvent_t start_after = nullptr, sized = nullptr;
  bool published = false;       // rh_k_publish ran: hctrl holds the compact layout (summed null counts) behind a token
  uint32_t token = 0;
  uint64_t o_flag_h = 0;

  rh_decode_call(rh_schema* s_, const uint8_t* data, const uint64_t* offs, uint64_t dl, uint64_t n_, uint64_t nc, const rh_opts* o,
               bool stats, const ChunkGeo* g, rh_device_result& res)
      : s(s_), cs(*s_->cs), d_data(data), d_offsets(offs), data_len(dl), n(n_), num_chunks(nc), opts(o ? *o : default_opts()),
        want_stats(stats), r(res) {
    std::memset(&st, 0, sizeof st);
    if (g) { geo_v = *g; geo = &geo_v; range_of_host_call = true; }
    opts.devices = nullptr; opts.n_devices = 0; opts.device_stats = nullptr; opts.ready = nullptr; opts.gathered = nullptr;   // (not used below; never dangling)
  }

  void check_bad(const uint8_t* h) {
    unsigned long long fb = *(const unsigned long long*)h;
    if (!fb) return;
    const uint64_t rec = ~fb;
    uint64_t c = r.sz ? std::min<uint64_t>(rec / r.sz, k - 1) : 0;
    uint64_t b = c * bpc64 + (rec - c * r.sz) / tile;
    rh::ErrInfo ei;
    HIPCHK(hipMemcpy(&ei, P.errinfo + b, sizeof ei, hipMemcpyDeviceToHost));
    throw DecodeError(format_error(ei));
  }

  // host statement of the layout (same rule, same table order as rh_k_layout): fills the result's tables
  void layout_host() {
    r.data_bytes = totals;
    for (auto t : totals)
      if (t > 0x7FFFFFFFull) {
        count(RH_CTR_OFFSET32_ERRORS);
        throw DecodeError("offset overflow: a chunk's column exceeds the 2^31-1 limit of 32-bit Arrow offsets");
      }
    if (sk)
      for (int d = 1; d < cs.ndom; d++)
        for (uint32_t c = 0; c < k; c++)
          if (totals[(size_t)(d - 1) * k + c] >= narrow_rows) throw NeedWideIndex();
    r.fill_tables();
    exact = r.output_bytes;
  }

  void launch_tail(bool offsets_done) {     // k_init + k_emit through the device tables at dtab
    if (nbuf > 0 && (child_bitmaps || !offsets_done) &&
        rh_launch_init(P.bufptr, d_sizes, dp->desc, (uint32_t)nbuf, k, P.first_bad, stream)) throw HipError("k_init launch failed");
    if (n > 0) {
      emit_lds = lds_bytes;
      if (sk ? launch_module(sk->emit_fn, P, nblocks, (uint32_t)tile, emit_lds, stream, ev.at(3), ev.at(4))
             : rh_launch_emit(&P, emit_lds, stream, ev.at(3), ev.at(4)))
        throw HipError("k_emit launch failed");
    } else {
      ev.rec(3, stream);
      ev.rec(4, stream);
    }
  }

  void exact_tail() {      // totals are on the host: exactly sized arena, tables from the host
    ctrl->b.clean = false;
    published = false;       // (the raw device layout is copied back below)
    layout_host();
    r.arena = Lease(dev_pool(), r.arena_bytes, device);
    Lease htab(pin_pool(), tab_bytes, device);
    void** hptr = (void**)htab.ptr();
    uint64_t* hsz = (uint64_t*)(htab.ptr() + (uint64_t)std::max(nbuf, 1) * k * 8);
    for (uint32_t c = 0; c < k; c++)
      for (int b = 0; b < nbuf; b++) {   // device tables are [chunk][buf]
        hptr[(size_t)c * nbuf + b] = r.arena.ptr() + r.buf_off[(size_t)b * k + c];
        hsz[(size_t)c * nbuf + b] = r.buf_size[(size_t)b * k + c];
      }
    HIPCHK(hipMemcpyAsync(dtab.ptr(), htab.ptr(), tab_bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(ctrl->ptr() + 8, 0, 8, stream));    // clear the layout flag (and the ticket) of a refused optimistic attempt
    launch_tail(false);
    // (not the totals: the host has them, and rh_k_publish may have zeroed the device copy of a refused attempt)
    HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), o_tot, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(hctrl.ptr() + o_null, ctrl->ptr() + o_null, ctrl_bytes - o_null, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));      // also keeps htab alive until the table copy is done
    check_bad(hctrl.ptr());
  }

  // The single-pass form (spec_body.h spec_fused): k_layout over per-column CAPACITIES from the schema's history, then ONE
  // kernel that sizes, scans across tiles (look-back) and emits, then rh_k_publish.  Returns false when the call does not
  // qualify (no history yet, generic kernels, knobs) -- the two-pass submission follows then.
  bool try_single(bool two_sync, long ratio_hook) {
    const bool on = (opts.flags & RH_SINGLE_PASS) != 0 || env_long("RUHVRO_HIP_SINGLE_PASS", kSinglePassDefault, 0, 1) != 0;
    // (device-resident calls only: a host call is bound by the PCIe link, and its D2H copy would carry the capacity slack)
    if (!on || range_of_host_call || (opts.flags & (RH_INTERNAL_TWO_PASS | RH_TWO_PASS)) || !sk || K <= 0 || K > 64 || n == 0 || two_sync ||
        ratio_hook >= 0)
      return false;
    if (n_entries > (1u << 16)) return false;
    if (8ull * K * k > 4ull * K * nblocks) return false;      // the capacities travel in the workspace's blocksum area (below)
    // the single-pass kernel is its own code object, compiled when a call first asks for it (in the background unless the
    // caller insists on specialised kernels): until it is there the call takes the two-pass form
    hipFunction_t fused_fn = sk->fused_fn.load(std::memory_order_acquire);
    if (!fused_fn) {
      if (sk->fused_dead) return false;
      fused_fn = spec_kernel(s, device, compile_policy(opts.flags & 3, n), false, true).fused_fn.load(std::memory_order_acquire);
      if (!fused_fn) return false;
    }
    std::vector<double> per_row;
    {
      std::lock_guard<std::mutex> g(s->mu);
      if ((int)s->per_row.size() != K) return false;
      if (s->single_cooldown > 0) { s->single_cooldown--; return false; }
      per_row = s->per_row;
    }
    // capacities: what the last call needed per row, + 4 % + a pad that covers a short chunk's noise; never more than the
    // 32-bit limits the kernels index with (a column that needs more overflows its capacity -> two-pass -> the usual errors)
    // (RUHVRO_HIP_SINGLE_SLACK_PERMILLE: knob / test hook -- below 1000 the capacities are smaller than what the last call
    //  needed, which forces the LF_CAPACITY fail-over to the two-pass form)
    const double slack = (double)env_long("RUHVRO_HIP_SINGLE_SLACK_PERMILLE", 1040, 1, 4000) / 1000.0;
    caps.assign((size_t)K * k, 0);
    for (int kk = 0; kk < K; kk++)
      for (uint32_t c = 0; c < k; c++) {
        const uint64_t rows_c = c == k - 1 ? r.rows_last : r.sz;
        uint64_t cap = (uint64_t)(per_row[(size_t)kk] * (double)rows_c * slack) + (slack >= 1.0 ? 4096 : 0);
        uint64_t lim = 0x7FFFFFFFull;
        if (kk < cs.ndom - 1) lim = std::min<uint64_t>(lim, narrow_rows - 1);       // a child row domain
        caps[(size_t)kk * k + c] = std::min(cap, lim);
      }
    {   // arena bytes of that layout (the rule of fill_tables / rh_k_layout)
      uint64_t off = 0;
      for (uint32_t c = 0; c < k; c++)
        for (int b = 0; b < nbuf; b++) {
          const rh::BufDesc& d = cs.bufs[b];
          const uint64_t rows0 = c == k - 1 ? r.rows_last : r.sz;
          const uint64_t rows = d.dom == 0 ? rows0 : caps[(size_t)(d.dom - 1) * k + c];
          off += rh::buf_slot_bytes(rh::buf_bytes(d.kind, rows, d.kind == rh::BK_DATA ? caps[(size_t)d.counter * k + c] : 0, nullptr, (uint32_t)d.counter));
        }
      arena_cap = std::max<uint64_t>(off, kAlign);
    }
    count(RH_CTR_SINGLE_PASS_CALLS);
    count(RH_CTR_FUSED_CALLS);             // (a single stream submission too)
    single = true; fused = true;
    r.arena = Lease(dev_pool(), arena_cap, device);
    lookback = Lease(dev_pool(), std::max<uint64_t>(8ull * K * nblocks, kAlign), device);
    HIPCHK(hipMemsetAsync(lookback.ptr(), 0, 8ull * K * nblocks, stream));
    // the capacities go to the device behind the leading words of the workspace's blocksum area (unused on this path)
    hcaps = Lease(pin_pool(), 8ull * K * k, device);
    std::memcpy(hcaps.ptr(), caps.data(), 8ull * K * k);
    uint64_t* d_caps = (uint64_t*)P.blocksum;
    HIPCHK(hipMemcpyAsync(d_caps, hcaps.ptr(), 8ull * K * k, hipMemcpyHostToDevice, stream));
    P.lookback = (unsigned long long*)lookback.ptr();
    P.caps = d_caps;
    rh::LParams LP;
    std::memset(&LP, 0, sizeof LP);
    LP.totals = d_caps; LP.desc = dp->desc; LP.sz = r.sz; LP.rows_last = r.rows_last; LP.n = n; LP.k = k;
    LP.nbuf = nbuf; LP.K = K; LP.ndom = cs.ndom; LP.arena = r.arena.ptr(); LP.capacity = r.arena.b.size;
    LP.bufptr = (void**)dtab.ptr(); LP.bufsize = d_sizes; LP.ctrl = P.first_bad; LP.narrow = 0;
    LP.narrow_rows = narrow_rows;
    if (rh_launch_layout(&LP, stream)) throw HipError("k_layout launch failed");
    if (nbuf > 0 && child_bitmaps && rh_launch_init(P.bufptr, d_sizes, dp->desc, (uint32_t)nbuf, k, P.first_bad, stream))
      throw HipError("k_init launch failed");
    const uint64_t tiles_max = std::max<uint64_t>((r.sz + tile - 1) / tile, (r.rows_last + tile - 1) / tile);
    emit_lds = lds_bytes;
    if (launch_module(fused_fn, P, (uint32_t)(tiles_max * k), (uint32_t)tile, emit_lds, stream, ev.at(3), ev.at(4)))
      throw HipError("k_fused launch failed");
    basis = (double)payload + 64.0 * (double)n;
    void* hdev = nullptr;
    if (hipHostGetDevicePointer(&hdev, hctrl.ptr(), 0) == hipSuccess && hdev) {
      static std::atomic<uint32_t> next_token{0x40000001u};
      token = next_token.fetch_add(1);
      if (token == 0) token = next_token.fetch_add(1);
      o_flag_h = align_up(o_null + 4ull * nnodes * k, 8);
      *(volatile uint32_t*)(hctrl.ptr() + o_flag_h) = 0;
      if (rh_launch_publish(ctrl->ptr(), hdev, (uint32_t)(o_null / 4), (uint32_t)(nnodes * (int)k), (uint32_t)(o_flag_h / 4), token, null_slots, stream))
        throw HipError("k_publish launch failed");
      ctrl->b.clean = true;
      published = true;
    } else {
      (void)hipGetLastError();
      HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
      if (async) done.record(device, stream);
    }
    return true;
  }

  void enqueue() {
    Range rk("ruhvro_hip:decode_device (k_size, k_scan, k_layout, k_init, k_emit)");
    if (opts.device >= 0) { HIPCHK(hipSetDevice(opts.device)); device = opts.device; }
    else HIPCHK(hipGetDevice(&device));
    stream = (hipStream_t)opts.stream;
    if ((uintptr_t)d_data & 15) throw std::invalid_argument("device payload pointer must be 16-byte aligned");

    {   // the generic kernels' dynamic-LDS limit is a per-device function attribute: set it once per device
      static std::mutex lds_mu;
      static std::vector<int> lds_done;
      std::lock_guard<std::mutex> g(lds_mu);
      if (std::find(lds_done.begin(), lds_done.end(), device) == lds_done.end()) {
        if (rh_set_max_lds(160 * 1024) != 0) throw HipError("cannot raise the dynamic LDS limit of the decode kernels");
        lds_done.push_back(device);
      }
    }

    r.cs = &cs;
    r.device = device;
    r.n = n;
    if (!geo && opts.chunk_rows) {   // a range of a larger call's chunks (one process per GPU): rh_opts.chunk_rows
      if (num_chunks < 1 || num_chunks > 0xFFFFFFFFull || (num_chunks - 1) > n / opts.chunk_rows ||
          (n > 0 && n == (num_chunks - 1) * opts.chunk_rows && num_chunks > 1))
        throw std::invalid_argument("chunk_rows: the n records do not make num_chunks chunks of chunk_rows rows (the last one takes the rest)");
      geo_v.k = (uint32_t)num_chunks;
      geo_v.sz = opts.chunk_rows;
      geo_v.rows_last = n - (num_chunks - 1) * opts.chunk_rows;
      geo_v.payload_bytes = data_len;
      geo = &geo_v;
    }
    k = geo ? geo->k : rh_clamp_chunks(n, num_chunks);
    r.k = k;
    r.sz = geo ? geo->sz : n / k;
    r.rows_last = geo ? geo->rows_last : n - (uint64_t)(k - 1) * r.sz;
    K = cs.K; nnodes = (int)cs.nodes.size(); nbuf = (int)cs.bufs.size();
    dp = &device_program(s, device);

    // kernel form: schema-specialised (compiled once per schema, cached) or the generic interpreter
    const int mode = opts.flags & 3;
    // the specialised kernels address every chunk buffer with 32-bit byte offsets
    // (every chunk buffer below 4 GiB: at most max_row_bytes per row -- 16 unless the schema has a wider fixed)
    // (RUHVRO_HIP_NARROW_ROWS: test hook that lowers the bound so that small inputs take the wide-index fallback)
    narrow_rows = (uint64_t)env_long("RUHVRO_HIP_NARROW_ROWS",
                                     (long)std::min<uint64_t>(1ull << 28, (1ull << 32) / std::max<uint32_t>(cs.max_row_bytes, 16)), 1, 1l << 28);
    const bool narrow_ok = std::max(r.sz, r.rows_last) < narrow_rows;
    if (mode != RH_KERNEL_GENERIC && n > 0 && narrow_ok) {
      const SpecKernel& k0 = spec_kernel(s, device, compile_policy(mode, n));
      if (k0.ok) sk = &k0;
      else if (mode == RH_KERNEL_SPECIALIZED) throw HipError("specialised kernel unavailable: " + k0.why);
    }
    tile = sk ? (uint64_t)rh::spec_tile_records() : (uint64_t)rh::kBlock;   // records per workgroup
    bpc64 = std::max<uint64_t>((r.sz + tile - 1) / tile, 1);
    const uint64_t nblocks64 = n == 0 ? 0 : (uint64_t)(k - 1) * bpc64 + (r.rows_last + tile - 1) / tile;
    if (nblocks64 > 0x7FFFFFFFull / std::max(K, 1)) throw std::invalid_argument("too many records for one call");
    nblocks = (uint32_t)nblocks64;

    // ---- control block: [first_bad u64 | layout flag, ticket | arena bytes | pad][totals u64 K*k][nullcount u32 nnodes*k*null_slots]
    //      workspace: errinfo | blocksum | blockbase | tileflag | lanecnt
    o_tot = 32;      // control words first (program.h): first_bad, layout flag, arena bytes used
    o_tick = o_tot + 8ull * K * k;                    // [k] tile tickets of the single-pass form (zero like the rest of the block)
    o_null = align_up(o_tick + 4ull * k, 16);
    null_slots = rh::null_slots_for(k);
    ctrl_bytes = align_up(o_null + 4ull * nnodes * k * null_slots, kAlign);
    const uint64_t o_err = 0;       // the rest lives in the workspace (needs no zeroing)
    const uint64_t o_bsum = align_up(o_err + sizeof(rh::ErrInfo) * (uint64_t)nblocks, kAlign);
    const uint64_t o_bbase = align_up(o_bsum + 4ull * K * nblocks, kAlign);
    const uint64_t o_flag = align_up(o_bbase + 4ull * K * nblocks, kAlign);
    const uint64_t o_lcnt = align_up(o_flag + (sk ? 4ull * nblocks : 0), kAlign);
    const uint64_t ws_bytes = align_up(o_lcnt + (sk ? 4ull * ((K + 1) / 2) * nblocks * tile : 0), kAlign);
    hp.mark("setup");
    ws = Lease(dev_pool(), ws_bytes, device);
    hctrl = Lease(pin_pool(), ctrl_bytes, device);
    ctrl.reset(new CtrlLease(ctrl_bytes, device, stream));        // all zero (CtrlPool)
    hp.mark("leases");

    std::memset(&P, 0, sizeof P);
    P.data = d_data; P.offsets = d_offsets; P.data_len = data_len;
    P.n = n; P.sz = r.sz; P.rows_last = r.rows_last; P.k = k; P.bpc = (uint32_t)bpc64; P.nblocks = nblocks;
    P.prog = dp->prog; P.sym_off = dp->sym_off; P.sym_data = dp->sym_data;
    P.nops = (int)cs.prog.size(); P.K = K; P.ndom = cs.ndom; P.nnodes = nnodes; P.list_depth = cs.list_depth;
    P.nbuf = nbuf; P.cnt_databuf = dp->cnt_databuf;
    P.first_bad = (unsigned long long*)ctrl->ptr();
    P.nullcount = (uint32_t*)(ctrl->ptr() + o_null);
    P.null_slots = null_slots;
    P.totals = (uint64_t*)(ctrl->ptr() + o_tot);
    P.tickets = (uint32_t*)(ctrl->ptr() + o_tick);
    P.errinfo = (rh::ErrInfo*)(ws.ptr() + o_err);
    P.blocksum = (uint32_t*)(ws.ptr() + o_bsum);
    P.blockbase = (uint32_t*)(ws.ptr() + o_bbase);
    P.tileflag = (uint32_t*)(ws.ptr() + o_flag);
    P.lanecnt = (uint32_t*)(ws.ptr() + o_lcnt);

    // LDS: fixed part + input window sized from the mean record length (falls back to global reads
    // for workgroups whose 256 records do not fit)
    const uint32_t lds_fixed = (sk ? rh::spec_lds_fixed_words_host(K, nnodes, (int)(tile / 64), rh::child_bitmap_count(cs), rh::dense_list_count(cs), rh::dom0_bitmap_count(cs)) * 4 : rh_lds_fixed_bytes(K, cs.list_depth, nnodes, nbuf)) + 16;   // + window slack
    payload = geo ? geo->payload_bytes : data_len;
    const uint64_t avg = n ? payload / n + 1 : 16;
    // (tuning / test knobs, read per call: RUHVRO_HIP_WIN_PCT, RUHVRO_HIP_WIN_PAD)
    const uint64_t win_pct = (uint64_t)env_long("RUHVRO_HIP_WIN_PCT", 115, 100, 400);
    const uint64_t win_pad = (uint64_t)env_long("RUHVRO_HIP_WIN_PAD", 2048, 0, 65536);
    uint64_t win = align_up(avg * tile * win_pct / 100 + win_pad * tile / rh::kBlock, 16);
    win = std::max<uint64_t>(win, 8192 * tile / rh::kBlock);
    const uint64_t lds_cap = 160 * 1024 - 512;
    if (lds_fixed + 4096 > lds_cap) throw rh::SchemaError("schema needs more LDS than a CDNA4 workgroup has");
    win = std::min<uint64_t>(win, std::min<uint64_t>((lds_cap - lds_fixed) & ~15ull, 96 * 1024));
    // Occupancy steps: a CU's 160 KB hold N workgroups of at most 160 KB / N each.  A window that puts the workgroup just
    // above a step costs a whole workgroup per CU (a quarter of the resident waves at N = 4) for a few hundred bytes of
    // slack, so it gives that slack up as long as a smaller margin (6 % + 1 KB over the mean tile) is left.
    if (win_pct == 115 && win_pad == 2048) {       // (not when a test / sweep sets the window by hand)
      const uint64_t min_win = align_up(avg * tile * 106 / 100 + 1024 * tile / rh::kBlock, 16);
      for (uint64_t nwg = 4; nwg >= 2; nwg--) {        // (4: what the emit kernel's registers allow at most)
        const uint64_t step = (160 * 1024 / nwg) & ~511ull;
        if (lds_fixed + win > step && step > lds_fixed && step - lds_fixed >= min_win) { win = (step - lds_fixed) & ~15ull; break; }
      }
    }
    P.win_bytes = (uint32_t)win;
    lds_bytes = lds_fixed + (uint32_t)win;
    // optional in-kernel phase timing of the specialised kernels (RUHVRO_HIP_PROFILE=1)
    static const bool profile_env = [] { const char* e = std::getenv("RUHVRO_HIP_PROFILE"); return e && *e && *e != '0'; }();
    profile = profile_env;
    if (profile && sk) {
      prof_buf = Lease(dev_pool(), 64 * 32 * 8, device);
      HIPCHK(hipMemsetAsync(prof_buf.ptr(), 0, 64 * 32 * 8, stream));
      P.prof = (unsigned long long*)prof_buf.ptr();
    }

    if (want_stats) ev.init();
    hp.mark("events");

    // ---- the launch sequence.  With a size history for this schema the whole call is ONE stream submission:
    //   k_size -> k_scan -> k_layout (exact arena layout on the device, program.h LParams) -> k_init -> k_emit -> one D2H
    // of the control words.  The arena is reserved up front from the history; when it turns out too small (the data
    // changed character), the layout kernel says so, init/emit return at once, and the host re-runs the tail with an
    // exactly sized arena -- which is also what the first call of a schema does.
    totals.assign((size_t)K * k, 0);
    n_entries = (uint64_t)k * std::max(nbuf, 0);
    tab_bytes = align_up((uint64_t)std::max(nbuf, 1) * k * 16, kAlign);
    dtab = Lease(dev_pool(), tab_bytes, device);
    d_sizes = (uint64_t*)(dtab.ptr() + (uint64_t)std::max(nbuf, 1) * k * 8);
    P.bufptr = (void* const*)dtab.ptr();
    for (const rh::BufDesc& d : cs.bufs) child_bitmaps = child_bitmaps || (d.kind == rh::BK_BITMAP && d.dom != 0);   // built with atomics on zeroed words

    const bool two_sync = env_long("RUHVRO_HIP_TWO_SYNC", 0, 0, 1) != 0;
    // (RUHVRO_HIP_ARENA_PERMILLE: test hook, the arena is reserved as if the schema's history said that many output
    //  bytes per 1000 input bytes -- a small value forces the LF_CAPACITY retry)
    const long ratio_hook = env_long("RUHVRO_HIP_ARENA_PERMILLE", -1, 0, 1000000);
    const double ratio = ratio_hook >= 0 ? std::max(1e-9, ratio_hook / 1000.0) : s->arena_ratio.load();
    fused = n > 0 && ratio > 0 && !two_sync && n_entries <= (1u << 16);
    // stage timings (rh_stats) come from the kernels' own start / stop timestamps: e0..e1 = k_size, e5..e2 = k_scan,
    // e3..e4 = k_emit
    if (start_after) HIPCHK(hipStreamWaitEvent(stream, start_after, 0));
    if (try_single(two_sync, ratio_hook)) return;
    timed_size = n > 0 && K > 0;
    if (timed_size) {
      if (sk ? launch_module(sk->size_fn, P, nblocks, (uint32_t)tile, lds_bytes, stream, ev.at(0), ev.at(1))
             : rh_launch_size(&P, lds_bytes, stream, ev.at(0), ev.at(1)))
        throw HipError("k_size launch failed");
      if (sized) HIPCHK(hipEventRecord(sized, stream));
      // (the single-submission path scans and lays the arena out in ONE launch, below)
      if (!fused && rh_launch_scan(&P, stream, ev.at(5), ev.at(2))) throw HipError("k_scan launch failed");
    } else {
      // no size pass (no variable-length output): nobody classified the tiles, so the emit kernel walks all of them carefully
      P.all_careful = 1;
      if (sized) HIPCHK(hipEventRecord(sized, stream));
    }
    // RUHVRO_HIP_NO_TRUST=1 (debugging aid): the emit pass walks EVERY tile with its own bounds and anomaly checks instead of
    // trusting the size pass's verdict on the same bytes (walk.h RH_TRUST) -- what a caller that suspects its input buffers
    // change between the two passes of an RH_ASYNC call turns on; the GPU suite passes with it (tests/test_async_device.py)
    static const bool no_trust = env_long("RUHVRO_HIP_NO_TRUST", 0, 0, 1) != 0;
    if (no_trust) P.all_careful = 1;
    hp.mark("size+scan_launch");
    basis = (double)payload + 64.0 * (double)n;
    if (fused) {
      count(RH_CTR_FUSED_CALLS);
      const uint64_t capacity = align_up((uint64_t)(ratio * basis * 1.125) + n_entries * kAlign + (1u << 20), kAlign);
      r.arena = Lease(dev_pool(), capacity, device);
      rh::LParams LP;
      std::memset(&LP, 0, sizeof LP);
      LP.totals = P.totals; LP.desc = dp->desc; LP.sz = r.sz; LP.rows_last = r.rows_last; LP.n = n; LP.k = k;
      LP.nbuf = nbuf; LP.K = K; LP.ndom = cs.ndom; LP.arena = r.arena.ptr(); LP.capacity = r.arena.b.size;
      LP.bufptr = (void**)dtab.ptr(); LP.bufsize = d_sizes; LP.ctrl = P.first_bad; LP.narrow = sk ? 1u : 0u;
      LP.narrow_rows = narrow_rows;
      if (timed_size ? rh_launch_scan_layout(&P, &LP, stream, ev.at(5), ev.at(2)) : rh_launch_layout(&LP, stream))
        throw HipError("k_scan / k_layout launch failed");
      launch_tail(true);                           // the layout kernel wrote offsets[0] = 0 itself
      hp.mark("layout+emit_launch");
      // the control words go to the host from the call's last kernel, which also re-zeroes the block (rh_k_publish) and
      // writes a per-call token behind them: finish() spins on that word instead of waiting for a stream event
      void* hdev = nullptr;
      static const bool no_publish = env_long("RUHVRO_HIP_NO_PUBLISH", 0, 0, 1) != 0;
      if (!no_publish && hipHostGetDevicePointer(&hdev, hctrl.ptr(), 0) == hipSuccess && hdev) {
        static std::atomic<uint32_t> next_token{1};
        token = next_token.fetch_add(1);
        if (token == 0) token = next_token.fetch_add(1);
        o_flag_h = align_up(o_null + 4ull * nnodes * k, 8);                 // host layout: head | compact null counts | token
        *(volatile uint32_t*)(hctrl.ptr() + o_flag_h) = 0;
        if (rh_launch_publish(ctrl->ptr(), hdev, (uint32_t)(o_null / 4), (uint32_t)(nnodes * (int)k), (uint32_t)(o_flag_h / 4), token, null_slots, stream))
          throw HipError("k_publish launch failed");
        ctrl->b.clean = true;
        published = true;
      } else {
        (void)hipGetLastError();
        HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
        if (async) done.record(device, stream);
      }
      hp.mark("d2h_enqueue");
    }
  }

  void finish() {
    if (settled) return;
    settled = true;
    HIPCHK(hipSetDevice(device));
    if (fused) {
      if (published) {
        // spin on the token rh_k_publish stores last into this call's pinned block: this call only (later calls stay
        // queued behind it), no event in the stream, and sooner than a stream wait returns
        volatile uint32_t* flag = (volatile uint32_t*)(hctrl.ptr() + o_flag_h);
        for (uint32_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != token;) {
          if ((++spins & 0xFFFFu) == 0) {              // a failed launch or a fault must not hang the caller
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {                     // everything on the stream is done: the token must be there
              if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != token) throw HipError("rh_k_publish finished without publishing its token");
              break;
            }
            if (q != hipErrorNotReady) throw HipError(std::string("stream failed while waiting for a decode call: ") + hipGetErrorString(q));
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      } else if (done.e) {
        done.wait();                               // this call only: later calls stay queued behind it
      } else {
        HIPCHK(hipStreamSynchronize(stream));
      }
      hp.mark("sync");
      check_bad(hctrl.ptr());
      if (K > 0) std::memcpy(totals.data(), hctrl.ptr() + o_tot, 8ull * K * k);
      const uint32_t lflag = *(const uint32_t*)(hctrl.ptr() + 8);
      if (single) {
        ctrl->b.clean = published;
        if (lflag) {            // a column outgrew its capacity (or the capacity layout was refused): the two-pass path decides
          // (not latched when the capacities were shrunk by the test hook)
          if ((lflag & rh::LF_CAPACITY) && env_long("RUHVRO_HIP_SINGLE_SLACK_PERMILLE", 1040, 1, 4000) >= 1000) {
            std::lock_guard<std::mutex> g(s->mu);
            s->single_backoff = std::min<uint32_t>(1024, std::max<uint32_t>(8, s->single_backoff * 2));
            s->single_cooldown = s->single_backoff;
          }
          count(RH_CTR_SINGLE_PASS_FAILOVERS);
          r.arena.release();
          throw NeedTwoPass();
        }
        r.data_bytes = totals;
        r.layout_bytes = caps;
        r.arena_bytes = arena_cap;
        { std::lock_guard<std::mutex> g(s->mu); s->single_backoff = 0; }
      } else if (lflag & rh::LF_CAPACITY) {
        count(RH_CTR_CAPACITY_RETRIES);
        r.arena.release();
        exact_tail();                              // (throws the offset-overflow / wide-index cases itself)
      } else if (lflag || want_stats) {
        layout_host();                             // throws for LF_OFFSET32 / LF_NEED_WIDE: same tests on the same totals
        if (lflag) throw HipError("internal error: layout kernel and host disagree");
        if (r.arena_bytes != std::max<uint64_t>(*(const uint64_t*)(hctrl.ptr() + 16), kAlign)) throw HipError("internal error: device and host arena layouts differ");
      } else {
        // the device laid the arena out and accepted it: the host's tables (same rule, same totals) wait for their first
        // reader (rh_device_result::tables) -- a caller that only hands the device buffers on never pays for them
        r.data_bytes = totals;
        r.arena_bytes = std::max<uint64_t>(*(const uint64_t*)(hctrl.ptr() + 16), kAlign);
      }
    } else {
      count(RH_CTR_TWO_SYNC_CALLS);
      if (n > 0 && K > 0) {
        HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        check_bad(hctrl.ptr());
        std::memcpy(totals.data(), hctrl.ptr() + o_tot, 8ull * K * k);
      }
      exact_tail();
    }
    if (n > 0 && basis > 0 && !single) {
      const double slots = (double)n_entries * (double)kAlign;
      s->arena_ratio.store(std::max(0.0, (double)r.arena_bytes - slots) / basis + 1e-9);
    }
    if (n > 0 && K > 0 && (int)totals.size() == K * (int)k) {       // per-row need of every counter's column (single-pass capacities)
      std::vector<double> pr((size_t)K, 0.0);
      for (int kk = 0; kk < K; kk++)
        for (uint32_t c = 0; c < k; c++) {
          const uint64_t rows_c = c == k - 1 ? r.rows_last : r.sz;
          if (rows_c) pr[(size_t)kk] = std::max(pr[(size_t)kk], (double)totals[(size_t)kk * k + c] / (double)rows_c);
        }
      std::lock_guard<std::mutex> g(s->mu);
      s->per_row = std::move(pr);
    }
    r.nullcount.assign((size_t)nnodes * k, 0);
    if (published) {           // rh_k_publish summed the slots: one word per (node, chunk)
      std::memcpy(r.nullcount.data(), hctrl.ptr() + o_null, 4ull * nnodes * k);
    } else {
      const uint32_t* slots = (const uint32_t*)(hctrl.ptr() + o_null);      // [nnodes][k][null_slots] (program.h)
      for (size_t e = 0; e < (size_t)nnodes * k; e++) {
        uint32_t sum = 0;
        for (uint32_t sl = 0; sl < null_slots; sl++) sum += slots[e * null_slots + sl];
        r.nullcount[e] = sum;
      }
    }
    hp.mark("host_layout");

    if (profile && sk) {
      unsigned long long hr[64 * 32], h[32] = {0};
      HIPCHK(hipMemcpy(hr, prof_buf.ptr(), sizeof hr, hipMemcpyDeviceToHost));
      for (int r0 = 0; r0 < 64; r0++)
        for (int i = 0; i < 32; i++) h[i] += hr[r0 * 32 + i];
      const double waves = (double)nblocks * 4;
      static const char* names2[] = {"offsets", "stage+barrier", "lane_init", "walk1", "scan", "barrier", "layout", "walk2",
                                     "errors+barrier", "flush"};
      static const char* names1[] = {"ticket+zero+barrier", "offsets+stage+barrier", "size_walk", "wave_scan", "barrier", "lookback(wave0)",
                                     "barrier", "prefix", "emit_walk", "errors+flush"};
      const char* const* names = single ? names1 : names2;
      std::fprintf(stderr, "[ruhvro_hip profile] %s cycles/wave:", single ? "single-pass" : "emit");
      for (int i = 0; i < 10; i++) std::fprintf(stderr, " %s=%.0f", names[i], h[i] / waves);
      std::fprintf(stderr, "\n[ruhvro_hip profile] size cycles/wave: stage+barrier=%.0f init=%.0f walk=%.0f tail=%.0f | kernels ms: size=%.3f emit=%.3f\n",
                   h[16] / waves, h[17] / waves, h[18] / waves, h[19] / waves, ev.ms(0, 1), ev.ms(3, 4));
    }
    if (want_stats) {
      st.records = n;
      st.input_bytes = payload;
      st.output_bytes = exact;
      st.chunks = k;
      st.blocks = nblocks;
      st.size_kernel_ms = (timed_size && !single) ? ev.ms(0, 1) : 0.f;
      st.scan_kernel_ms = (timed_size && !single) ? ev.ms(5, 2) : 0.f;
      st.emit_kernel_ms = n > 0 ? ev.ms(3, 4) : 0.f;
      st.specialized = sk ? 1 : 0;
      st.lds_bytes = emit_lds;
    }
    // the call's scratch goes back to the pools now (the control block is zeroed on its stream, CtrlPool)
    ws.release(); dtab.release(); hctrl.release(); prof_buf.release(); lookback.release(); hcaps.release(); ctrl.reset();
  }

  // a call that failed (or is abandoned) must not hand its blocks back while the GPU may still be using them
  void drain() noexcept {
    if (stream || device >= 0) { (void)hipSetDevice(device); (void)hipStreamSynchronize(stream); }
  }
};

rh_device_result::rh_device_result() { std::memset(&st, 0, sizeof st); }
rh_device_result::~rh_device_result() {
  if (pending) {                 // freed without a wait: the GPU may still be writing into the blocks this result owns
    pending->drain();
    pending.reset();
  }
  parts.clear();                 // (each group drains its own stream)
  if (!join_events.empty()) {
    std::lock_guard<std::mutex> g(DoneEvent::mu());
    auto& v = DoneEvent::idle()[device];
    for (hipEvent_t e : join_events) {
      if (v.size() < 64) v.push_back(e);
      else (void)hipEventDestroy(e);
    }
  }
}

namespace {
typedef rh_decode_call DeviceDecode;

rh_device_result* decode_device_impl1(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len,
                                      uint64_t n, uint64_t num_chunks, const rh_opts* opts, rh_stats* stats, const ChunkGeo* geo) {
  auto res = std::make_unique<rh_device_result>();
  auto call = std::make_unique<DeviceDecode>(s, d_data, d_offsets, data_len, n, num_chunks, opts, stats != nullptr, geo, *res);
  const bool async = opts && (opts->flags & RH_ASYNC) && !geo;
  call->async = async;
  try {
    call->enqueue();
    if (async && call->fused) {            // everything is on the stream: settle later (rh_device_result_wait)
      res->pending = std::move(call);
      return res.release();
    }
    call->finish();
  } catch (...) {
    call->drain();
    throw;
  }
  if (stats) {
    const float pack = stats->pack_ms, h2d = stats->h2d_ms, d2h = stats->d2h_ms, tot = stats->total_ms;
    *stats = call->st;
    stats->pack_ms = pack; stats->h2d_ms = h2d; stats->d2h_ms = d2h; stats->total_ms = tot;
  }
  return res.release();
}


// ---------------------------------------------------------------------------
// In-call overlap: a large device-resident call deals its chunk GROUPS to internal streams.
//
// The reference runs one task per chunk (ruhvro/src/deserialize.rs:92-120); chunks are independent here too, and the two
// passes load different parts of a CU (the size pass is bound by VALU issue, the emit pass co-limited by the vector-memory
// path), so the size pass of one group running beside the emit pass of another fills issue slots that either kernel
// alone leaves empty (bench.py `overlapped` measured it between independent calls; this is the same inside ONE call).
// Group g = chunks [k*g/G, k*(g+1)/G) is a complete sub-call (size -> scan+layout -> emit -> publish, its own arena and
// control block) on its own stream; the caller's stream forks into the internal streams at the start of the call and
// joins them at its end, so the result is valid in stream order on rh_opts.stream exactly like an unsplit call's.
// Not taken when the caller asks for stage timings (kernels that share the chip have no per-kernel duration), on a
// schema's first call (no size history), for the generic kernels, or below RUHVRO_HIP_SPLIT_MIN records.
// RUHVRO_HIP_INTERNAL_STREAMS=G (1 = off) -- profiler passes run with 1.
// ---------------------------------------------------------------------------
void settle(rh_device_result* r);

hipEvent_t pooled_event(int device) {
  hipEvent_t e = nullptr;
  {
    std::lock_guard<std::mutex> g(DoneEvent::mu());
    auto& v = DoneEvent::idle()[device];
    if (!v.empty()) { e = v.back(); v.pop_back(); }
  }
  if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

// the internal streams that accompany one caller stream on one device (created on first use, kept for the process)
std::vector<hipStream_t> companion_streams(int device, hipStream_t caller, unsigned want) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::vector<hipStream_t>> all;
  std::lock_guard<std::mutex> g(mu);
  if (all.size() >= 64 && !all.count({device, caller})) return {};          // a caller that burns through streams: no split
  auto& v = all[{device, caller}];
  while (v.size() < want) {
    hipStream_t x = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    v.push_back(x);
  }
  return std::vector<hipStream_t>(v.begin(), v.begin() + want);
}

rh_device_result* decode_device_split(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len, uint64_t n,
                                      const rh_opts& opts, uint32_t k, uint64_t sz, uint64_t rows_last, unsigned NS) {
  int device = opts.device;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  hipStream_t caller = (hipStream_t)opts.stream;
  const std::vector<hipStream_t> extra = companion_streams(device, caller, NS - 1);
  if (extra.size() != NS - 1) return nullptr;
  // groups: G >= NS runs of whole chunks, dealt to the NS streams round-robin (RUHVRO_HIP_SPLIT_GROUPS, default = NS);
  // staggered (RUHVRO_HIP_SPLIT_STAGGER, default on): group g + 1's size pass starts when group g's has finished, so it
  // runs beside group g's EMIT pass (different bounds) instead of beside its size pass (the same bound)
  const unsigned G = (unsigned)std::min<uint64_t>(k, (uint64_t)std::max<long>(env_long("RUHVRO_HIP_SPLIT_GROUPS", 0, 0, 64), (long)NS));
  const bool stagger = env_long("RUHVRO_HIP_SPLIT_STAGGER", 1, 0, 1) != 0;
  hipEvent_t prev_sized = nullptr;
  auto parent = std::make_unique<rh_device_result>();
  parent->cs = s->cs.get(); parent->device = device; parent->n = n; parent->k = k; parent->sz = sz; parent->rows_last = rows_last;
  const bool async = (opts.flags & RH_ASYNC) != 0;
  count(RH_CTR_SPLIT_CALLS);
  // fork: the internal streams start behind everything that is on the caller's stream now (the input buffers' producers)
  hipEvent_t fork = pooled_event(device);
  parent->join_events.push_back(fork);
  HIPCHK(hipEventRecord(fork, caller));
  for (hipStream_t x : extra) HIPCHK(hipStreamWaitEvent(x, fork, 0));
  for (unsigned g = 0; g < G; g++) {
    const uint32_t c0 = (uint32_t)((uint64_t)k * g / G), c1 = (uint32_t)((uint64_t)k * (g + 1) / G);
    const uint64_t r0 = (uint64_t)c0 * sz, r1 = c1 == k ? n : (uint64_t)c1 * sz;
    ChunkGeo geo;
    geo.k = c1 - c0; geo.sz = sz; geo.rows_last = c1 == k ? rows_last : sz;
    geo.payload_bytes = n ? (uint64_t)((double)data_len * (double)(r1 - r0) / (double)n) : 0;   // (the offsets live on the device)
    rh_opts o = opts;
    o.stream = g % NS == 0 ? (void*)caller : (void*)extra[g % NS - 1];
    o.device = device; o.chunk_rows = 0; o.flags &= ~RH_ASYNC;
    auto res = std::make_unique<rh_device_result>();
    auto call = std::make_unique<DeviceDecode>(s, d_data, d_offsets + r0, data_len, r1 - r0, (uint64_t)geo.k, &o, false, &geo, *res);
    call->async = true;
    if (stagger) {
      call->start_after = prev_sized;
      if (g + 1 < G) {
        prev_sized = pooled_event(device);
        parent->join_events.push_back(prev_sized);
        call->sized = prev_sized;
      }
    }
    try {
      call->enqueue();
      if (call->fused) {
        res->pending = std::move(call);
      } else {
        call->finish();
      }
    } catch (...) {
      call->drain();
      throw;                      // (the groups already enqueued drain in the parent's destructor)
    }
    parent->part_chunk0.push_back(c0);
    parent->parts.push_back(std::move(res));
  }
  // join: the caller's stream continues behind every group
  for (hipStream_t x : extra) {
    hipEvent_t e = pooled_event(device);
    parent->join_events.push_back(e);
    HIPCHK(hipEventRecord(e, x));
    HIPCHK(hipStreamWaitEvent(caller, e, 0));
  }
  if (!async) settle(parent.get());
  return parent.release();
}

// The host's half of an asynchronous call (RH_ASYNC): wait for the stream, check for a malformed record, retry with an
// exact arena if the reserved one was too small, fall back to the generic kernels if a child row domain needs 64-bit
// indexing.  Throws what the synchronous call would have thrown; a failed result stays failed.
void settle(rh_device_result* r) {
  if (r->fail) std::rethrow_exception(r->fail);
  if (!r->parts.empty()) {
    // groups are settled in chunk order: the first failure is the lowest failing group's, i.e. the lowest malformed
    // record of the call (the in-order join of deserialize.rs:115-119)
    try {
      for (auto& p : r->parts) settle(p.get());
    } catch (...) {
      r->fail = std::current_exception();
      throw;
    }
    return;
  }
  if (!r->pending) return;
  std::unique_ptr<DeviceDecode> call = std::move(r->pending);
  // the call again, synchronously, on another form: the two-pass form (a single-pass call that outgrew a capacity) or the
  // generic kernels (a child row domain beyond 32-bit indexing -- which the two-pass repeat may itself run into)
  auto rerun = [&](int add_flags, bool generic) {
    call->drain();
    if (generic) count(RH_CTR_WIDE_FALLBACKS);
    rh_opts o = call->opts;
    o.flags = generic ? ((o.flags & ~(3 | RH_ASYNC)) | RH_KERNEL_GENERIC) : ((o.flags & ~RH_ASYNC) | add_flags);
    rh_stats st2;
    std::memset(&st2, 0, sizeof st2);
    std::unique_ptr<rh_device_result> r2;
    try {
      r2.reset(decode_device_impl1(call->s, call->d_data, call->d_offsets, call->data_len, call->n, call->num_chunks, &o,
                                   call->want_stats ? &st2 : nullptr, call->geo));
    } catch (const NeedWideIndex&) {
      if (generic) throw;
      count(RH_CTR_WIDE_FALLBACKS);
      o.flags = (o.flags & ~3) | RH_KERNEL_GENERIC;
      r2.reset(decode_device_impl1(call->s, call->d_data, call->d_offsets, call->data_len, call->n, call->num_chunks, &o,
                                   call->want_stats ? &st2 : nullptr, call->geo));
    }
    call.reset();                                  // (its reference to *r ends here)
    r->arena = std::move(r2->arena);
    r->arena_bytes = r2->arena_bytes;
    r->buf_off = std::move(r2->buf_off); r->buf_size = std::move(r2->buf_size); r->dom_rows = std::move(r2->dom_rows);
    r->data_bytes = std::move(r2->data_bytes); r->nullcount = std::move(r2->nullcount); r->layout_bytes = std::move(r2->layout_bytes);
    r->output_bytes = r2->output_bytes; r->tables_done = r2->tables_done;
    r->k = r2->k; r->sz = r2->sz; r->rows_last = r2->rows_last;
    if (r2->has_stats || st2.records) { r->st = st2; r->has_stats = true; }
  };
  try {
    try {
      call->finish();
      if (call->want_stats) { r->st = call->st; r->has_stats = true; }
    } catch (const NeedTwoPass&) {
      rerun(RH_INTERNAL_TWO_PASS, false);
    } catch (const NeedWideIndex&) {
      rerun(0, true);
    }
  } catch (...) {
    if (call) call->drain();
    r->arena.release();
    r->fail = std::current_exception();
    throw;
  }
}

// Device range -> freshly owned host memory.  Large results land in pooled PINNED memory (the copy then runs at PCIe
// speed, 57 GB/s measured) as long as a cached block is free or the pinned memory lent to still-live results stays
// under a bound; a caller that keeps many results alive gets pageable memory instead of a fresh 0.15 ms/MB
// hipHostMalloc per call.
Slab* slab_from_device(const uint8_t* dptr, uint64_t bytes, int device, hipStream_t stream = nullptr) {
  Slab* slab = new Slab();
  try {
    if (bytes >= (1ull << 20)) {
      slab->pinned = pin_pool().try_get(bytes, device);
      const uint64_t bound = std::max<uint64_t>(4ull << 30, 2 * bytes);
      if (!slab->pinned.p && Slab::pinned_result_bytes().load() + bytes <= bound)
        slab->pinned = pin_pool().get(bytes, device);
      if (slab->pinned.p) {
        Slab::pinned_result_bytes().fetch_add(slab->pinned.size);
        slab->base = slab->pinned.p;
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

int to_host_impl(rh_device_result* r, ArrowArray* out_chunks, hipStream_t stream = nullptr) {
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
  Slab* slab = slab_from_device(r->arena.ptr(), r->arena_bytes, r->device, stream);
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

template <typename F>
int guarded(char** err, F&& f) {
  try {
    return f();
  } catch (const rh::SchemaError& e) {
    if (err) *err = dup_msg(e.what());
    return RH_ERR_SCHEMA;
  } catch (const DecodeError& e) {
    if (err) *err = dup_msg(e.what());
    return RH_ERR_DECODE;
  } catch (const ValueClassError& e) {
    if (err) *err = dup_msg(e.what());
    return RH_ERR_DECODE;
  } catch (const std::invalid_argument& e) {
    if (err) *err = dup_msg(e.what());
    return RH_ERR_ARGUMENT;
  } catch (const std::exception& e) {
    if (err) *err = dup_msg(e.what());
    return RH_ERR_RUNTIME;
  } catch (...) {            // (an engine-internal signal that no handler claimed must never take the process down)
    if (err) *err = dup_msg("internal error: unhandled engine signal");
    return RH_ERR_RUNTIME;
  }
}

void require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw HipError("no HIP device available: the ruhvro_hip engine has no CPU decode path");
}

// Lets the groups of a pipelined call through one PCIe direction in group order, one at a time: group g+1's H2D then
// runs while group g's kernels and D2H do, and the two directions of the link stay busy together.
struct Turnstile {
  std::mutex mu;
  std::condition_variable cv;
  uint32_t next = 0;
  void enter(uint32_t ticket) {
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] { return next == ticket; });
  }
  void leave() {
    { std::lock_guard<std::mutex> l(mu); next++; }
    cv.notify_all();
  }
  void finish(uint32_t ticket) {      // a group that never reached this gate (it failed earlier) must not hold up its successors
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] { return next >= ticket; });
    if (next == ticket) {
      next++;
      l.unlock();
      cv.notify_all();
    }
  }
};
struct TurnstilePass {      // RAII: a group that fails still lets the next one in
  Turnstile* t;
  bool in = false;
  TurnstilePass(Turnstile* ts, uint32_t ticket) : t(ts) { if (t) { t->enter(ticket); in = true; } }
  void done() { if (t && in) { t->leave(); in = false; } }
  ~TurnstilePass() { done(); }
};

// Where a call's records are: packed (one payload + n+1 absolute offsets: rh_decode_packed, what the reference builds
// at deserialize.rs:90) or one (pointer, length) slice per record (rh_decode: what src/lib.rs:29-33 extracts).
struct Source {
  const uint8_t* data = nullptr;
  const uint64_t* offsets = nullptr;
  const uint8_t* const* ptrs = nullptr;
  const uint64_t* lens = nullptr;
  bool slices() const { return ptrs != nullptr || (data == nullptr && offsets == nullptr); }
};

void run_threads(unsigned nt, const std::function<void(unsigned)>& f) {
  if (nt <= 1) { f(0); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back(f, t);
  for (auto& x : th) x.join();
}

// A pool of host threads that lives for one call: parallel_for hands out task indices to the workers and returns when
// all are done.  The gather of a pipelined call runs shard after shard on it (creating 2 x 32 threads per shard instead
// costs more than the gather itself).
class CallPool {
 public:
  explicit CallPool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) th_.emplace_back([this] { work(); });
  }
  ~CallPool() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
    cv_start_.notify_all();
    for (auto& t : th_) t.join();
  }
  unsigned workers() const { return (unsigned)th_.size(); }
  void parallel_for(unsigned ntasks, const std::function<void(unsigned)>& f) {     // one caller at a time
    if (ntasks == 0) return;
    std::unique_lock<std::mutex> l(mu_);
    fn_ = &f; ntasks_ = ntasks; next_ = 0; left_ = ntasks; gen_++;
    cv_start_.notify_all();
    cv_done_.wait(l, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_start_.wait(l, [&] { return stop_ || (gen_ != seen && next_ < ntasks_); });
      if (stop_) return;
      seen = gen_;
      while (fn_ && next_ < ntasks_) {
        const unsigned t = next_++;
        const std::function<void(unsigned)>* f = fn_;
        l.unlock();
        (*f)(t);
        l.lock();
        if (--left_ == 0) cv_done_.notify_all();
        if (gen_ != seen) break;          // (cannot happen before left_ == 0; kept for clarity)
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_start_, cv_done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned ntasks_ = 0, next_ = 0, left_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// The gather pool and the shard streams of a pipelined host call are KEPT between calls: starting 32 threads while the caller's
// extractor threads are busy took 2.5 ms of a 9 ms call (1M records through the Python surface, RUHVRO_HIP_TIMELINE,
// profiles/r04zc_*); a second concurrent host call gets a pool of its own (at most two idle ones are kept).  A forked child starts
// empty (the threads do not exist there): the cached objects are abandoned, never used.
struct HostCallCache {
  std::mutex mu;
  std::vector<std::unique_ptr<CallPool>> pools;
  std::vector<std::pair<int, hipStream_t>> streams;
  static HostCallCache& get() {
    static HostCallCache* c = [] {
      HostCallCache* x = new HostCallCache();      // (never destroyed: worker threads may outlive static destruction order)
      pthread_atfork(nullptr, nullptr, [] {
        HostCallCache& h = get();
        new (&h.mu) std::mutex();
        for (auto& p : h.pools) (void)p.release();
        h.pools.clear();
        h.streams.clear();
      });
      return x;
    }();
    return *c;
  }
  std::unique_ptr<CallPool> take_pool(unsigned workers) {
    {
      std::lock_guard<std::mutex> l(mu);
      for (size_t i = 0; i < pools.size(); i++)
        if (pools[i]->workers() == workers) {
          std::unique_ptr<CallPool> p = std::move(pools[i]);
          pools.erase(pools.begin() + (long)i);
          return p;
        }
    }
    return std::unique_ptr<CallPool>(new CallPool(workers));
  }
  void give_pool(std::unique_ptr<CallPool> p) {
    std::lock_guard<std::mutex> l(mu);
    if (pools.size() < 2) pools.push_back(std::move(p));
  }
  hipStream_t take_stream(int device) {
    {
      std::lock_guard<std::mutex> l(mu);
      for (size_t i = 0; i < streams.size(); i++)
        if (streams[i].first == device) {
          hipStream_t st = streams[i].second;
          streams.erase(streams.begin() + (long)i);
          return st;
        }
    }
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
  }
  void give_stream(int device, hipStream_t st, bool idle) {
    if (idle) {
      std::lock_guard<std::mutex> l(mu);
      if (streams.size() < 16) { streams.emplace_back(device, st); return; }
    }
    (void)hipStreamDestroy(st);
  }
};

// Record slices [r0, r0 + n) gathered into pooled PINNED memory together with their offsets, laid out exactly like the
// device staging buffer: [16 bytes lead][payload][pad to kAlign][u64 offsets n+1].
struct Gathered {
  Lease pin;
  uint64_t tot = 0, o_off = 0, total_bytes = 0;
  float pack_ms = 0.f;
  static constexpr uint64_t lead = 16;
  // staged range of a PACKED source (stage_packed): [lead_packed][payload lo..hi][pad][u64 absolute offsets n+1]
  bool packed = false;
  uint64_t lo = 0, hi = 0, lead_packed = 0;
};

// `par(ntasks, f)` runs f(0..ntasks-1) on host threads and returns when all are done
Gathered gather_slices(const Source& src, uint64_t r0, uint64_t n, int device, unsigned nt_in,
                       const std::function<void(unsigned, const std::function<void(unsigned)>&)>& par) {
  Range rg("ruhvro_hip:gather");
  Timer tp;
  Gathered g;
  const uint8_t* const* ptrs = src.ptrs + r0;
  const uint64_t* lens = src.lens + r0;
  // pass 1: byte totals per thread range; pass 2: offsets + bytes
  const unsigned nt = n >= 4096 ? std::max(1u, nt_in) : 1u;
  auto lo_of = [&](unsigned t) { return n * t / nt; };
  std::vector<uint64_t> part(nt + 1, 0);
  par(nt, [&](unsigned t) {
    uint64_t sum = 0;
    for (uint64_t i = lo_of(t); i < lo_of(t + 1); i++) sum += lens[i];
    part[t + 1] = sum;
  });
  for (unsigned t = 0; t < nt; t++) part[t + 1] += part[t];
  g.tot = part[nt];
  g.o_off = align_up(Gathered::lead + g.tot + 32, kAlign);
  g.total_bytes = g.o_off + 8 * (n + 1);
  g.pin = Lease(pin_pool(), g.total_bytes, device);
  uint8_t* hdst = g.pin.ptr() + Gathered::lead;
  uint64_t* hoff = (uint64_t*)(g.pin.ptr() + g.o_off);
  par(nt, [&](unsigned t) {
    uint64_t pos = part[t];
    for (uint64_t i = lo_of(t); i < lo_of(t + 1); i++) {
      hoff[i] = pos;
      std::memcpy(hdst + pos, ptrs[i], lens[i]);
      pos += lens[i];
    }
  });
  hoff[n] = g.tot;
  g.pack_ms = tp.ms();
  return g;
}

// rh_bench_gather (test hook): the two passes of gather_slices over rows [r0, r0 + n) into a destination the caller
// provides -- what bounds the HOST side of a g-GPU call, measurable without g GPUs.  Returns the payload bytes.
uint64_t gather_into(const uint8_t* const* ptrs, const uint64_t* lens, uint64_t n, unsigned nt_in, uint8_t* hdst, uint64_t* hoff) {
  const unsigned nt = n >= 4096 ? std::max(1u, nt_in) : 1u;
  auto lo_of = [&](unsigned t) { return n * t / nt; };
  std::vector<uint64_t> part(nt + 1, 0);
  run_threads(nt, [&](unsigned t) {
    uint64_t sum = 0;
    for (uint64_t i = lo_of(t); i < lo_of(t + 1); i++) sum += lens[i];
    part[t + 1] = sum;
  });
  for (unsigned t = 0; t < nt; t++) part[t + 1] += part[t];
  run_threads(nt, [&](unsigned t) {
    uint64_t pos = part[t];
    for (uint64_t i = lo_of(t); i < lo_of(t + 1); i++) {
      hoff[i] = pos;
      std::memcpy(hdst + pos, ptrs[i], lens[i]);
      pos += lens[i];
    }
  });
  hoff[n] = part[nt];
  return part[nt];
}

// Rows [r0, r0 + n) of a PACKED source in pageable memory, copied into pooled PINNED memory by the call's host
// threads in the layout of the device staging buffer (one H2D copy then takes the whole range).  The runtime stages a
// pageable H2D copy through its own bounce buffers on the calling thread, and such copies do not overlap with another
// stream's D2H (measured in round 1: 56.5 vs 55.0 ms pipelined vs not); staged here they are ordinary DMA from pinned
// memory, so a large rh_decode_packed call is pipelined like rh_decode -- group g+1's records go in while group g's
// Arrow buffers come out.
Gathered stage_packed_range(const Source& src, uint64_t r0, uint64_t n, int device, unsigned nt_in,
                            const std::function<void(unsigned, const std::function<void(unsigned)>&)>& par) {
  Range rg("ruhvro_hip:stage");
  Timer tp;
  Gathered g;
  g.packed = true;
  const uint64_t* offsets = src.offsets + r0;
  g.lo = offsets[0]; g.hi = offsets[n];
  g.tot = g.hi - g.lo;
  g.lead_packed = 16 + (g.lo & 15);
  g.o_off = align_up(g.lead_packed + g.tot + 32, kAlign);
  g.total_bytes = g.o_off + 8 * (n + 1);
  g.pin = Lease(pin_pool(), g.total_bytes, device);
  uint8_t* hdst = g.pin.ptr() + g.lead_packed;
  uint8_t* hoff = g.pin.ptr() + g.o_off;
  const uint64_t obytes = 8 * (n + 1);
  const unsigned nt = g.tot >= (4u << 20) ? std::max(1u, nt_in) : 1u;
  par(nt, [&](unsigned t) {
    const uint64_t a = g.tot * t / nt, b = g.tot * (t + 1) / nt;
    if (b > a) std::memcpy(hdst + a, src.data + g.lo + a, b - a);
    const uint64_t oa = obytes * t / nt & ~7ull, ob = t + 1 == nt ? obytes : (obytes * (t + 1) / nt & ~7ull);
    if (ob > oa) std::memcpy(hoff + oa, (const uint8_t*)offsets + oa, ob - oa);
  });
  g.pack_ms = tp.ms();
  return g;
}

// Rows [r0, r1) of the source: (gather +) H2D, the kernels, D2H -- all on `stream`.
// Slices are gathered into pooled PINNED memory together with their offsets, laid out exactly like the device
// staging buffer, so the range goes up in ONE copy (the reference's BinaryArray::from_vec, deserialize.rs:90, but
// per shard -- a later shard gathers while an earlier one is on the wire -- and straight into DMA-able memory).
void decode_range(rh_schema* s, const Source& src, uint64_t r0, uint64_t r1, uint64_t num_chunks, const ChunkGeo* geo_in,
                  const rh_opts* opts, int device, hipStream_t stream, ArrowArray* out_chunks, uint32_t* out_k,
                  rh_stats* stats, Turnstile* h2d_gate, Turnstile* d2h_gate, uint32_t ticket, unsigned pack_threads,
                  Gathered* pre = nullptr) {
  const uint64_t n = r1 - r0;
  rh_opts o = default_opts();
  o.device = device;
  o.flags = opts ? (opts->flags & 3) : 0;      // kernel form only: the host paths settle every device call themselves
  o.stream = (void*)stream;
  float h2d = 0.f, pack_ms = 0.f;
  Lease din, pin;
  const uint8_t* base = nullptr;
  const uint64_t* d_offsets = nullptr;
  uint64_t data_end = 0;
  ChunkGeo geo;
  if (geo_in) geo = *geo_in;
  if (src.slices()) {
    // a pipelined call gathered this shard already (in shard order, on the call's thread pool); else gather here
    Gathered own;
    if (!pre) {
      own = gather_slices(src, r0, n, device, pack_threads,
                          [](unsigned nt, const std::function<void(unsigned)>& f) { run_threads(nt, f); });
      pre = &own;
      if (opts && opts->gathered && !geo_in) __atomic_store_n(opts->gathered, r1, __ATOMIC_RELEASE);   // (the unpipelined call)
    }
    pin = std::move(pre->pin);
    pack_ms = pre->pack_ms;
    const uint64_t tot = pre->tot, lead = Gathered::lead, o_off = pre->o_off, total_bytes = pre->total_bytes;
    din = Lease(dev_pool(), total_bytes, device);
    {
      TurnstilePass pass(h2d_gate, ticket);
      Timeline::mark(ticket, "h2d begin");
      Range rh("ruhvro_hip:h2d");
      Timer th;
      HIPCHK(hipMemcpyAsync(din.ptr(), pin.ptr(), total_bytes, hipMemcpyHostToDevice, stream));
      if (h2d_gate || stats) HIPCHK(hipStreamSynchronize(stream));   // a gate orders the shards of one link; else the stream does
      h2d = th.ms();
      Timeline::mark(ticket, "h2d end");
    }
    base = din.ptr() + lead;
    d_offsets = (const uint64_t*)(din.ptr() + o_off);
    data_end = tot;
    geo.payload_bytes = tot;
  } else if (pre) {
    // a pipelined call staged this range in pinned memory already (stage_packed_range): one DMA copy
    pin = std::move(pre->pin);
    pack_ms = pre->pack_ms;
    din = Lease(dev_pool(), pre->total_bytes, device);
    {
      TurnstilePass pass(h2d_gate, ticket);
      Timeline::mark(ticket, "h2d begin");
      Range rh("ruhvro_hip:h2d");
      Timer th;
      HIPCHK(hipMemcpyAsync(din.ptr(), pin.ptr(), pre->total_bytes, hipMemcpyHostToDevice, stream));
      if (h2d_gate || stats) HIPCHK(hipStreamSynchronize(stream));
      h2d = th.ms();
      Timeline::mark(ticket, "h2d end");
    }
    base = din.ptr() + pre->lead_packed - pre->lo;      // absolute offsets, virtual base (see below)
    d_offsets = (const uint64_t*)(din.ptr() + pre->o_off);
    data_end = pre->hi;
    geo.payload_bytes = pre->tot;
  } else {
    const uint64_t* offsets = src.offsets + r0;
    const uint64_t lo = offsets[0], hi = offsets[n];
    // the kernels index the payload with the absolute offsets: hand them a (virtual) base such that base + lo is where
    // the range's first byte lands, congruent to lo modulo 16 so that the 16-byte window rows stay aligned
    const uint64_t lead = 16 + (lo & 15);
    const uint64_t o_off = align_up(lead + (hi - lo) + 32, kAlign);
    din = Lease(dev_pool(), o_off + 8 * (n + 1), device);
    {
      TurnstilePass pass(h2d_gate, ticket);
      Range rh("ruhvro_hip:h2d");
      Timer th;
      if (hi > lo) HIPCHK(hipMemcpyAsync(din.ptr() + lead, src.data + lo, hi - lo, hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(din.ptr() + o_off, offsets, 8 * (n + 1), hipMemcpyHostToDevice, stream));
      if (h2d_gate || stats) HIPCHK(hipStreamSynchronize(stream));
      h2d = th.ms();
    }
    base = din.ptr() + lead - lo;
    d_offsets = (const uint64_t*)(din.ptr() + o_off);
    data_end = hi;
    geo.payload_bytes = hi - lo;
  }
  std::unique_ptr<rh_device_result> r(decode_device_impl(s, base, d_offsets, data_end, n, num_chunks, &o, stats,
                                                         geo_in ? &geo : nullptr));
  pin.release();            // the staging copy is done (decode_device_impl synchronised the stream)
  Timeline::mark(ticket, "kernels end");
  float d2h = 0.f;
  {
    TurnstilePass pass(d2h_gate, ticket);
    Timeline::mark(ticket, "d2h begin");
    Range rd("ruhvro_hip:d2h+export");
    Timer td;
    to_host_impl(r.get(), out_chunks, stream);
    d2h = td.ms();
    Timeline::mark(ticket, "d2h end");
  }
  if (out_k) *out_k = r->k;
  if (stats) {
    stats->h2d_ms = h2d;
    stats->d2h_ms = d2h;
    stats->pack_ms = pack_ms;
  }
}

// Payload bytes from which a call is pipelined.  Measured on MI355X (profiles/r01h_pipeline_e2e.jsonl, 10M records,
// 1.2 GB in / 1.7 GB out): with the records in PINNED memory (rh_decode packs them there) the two PCIe directions
// overlap and H2D + kernels + D2H drop from 55 to 43 ms; from PAGEABLE memory (rh_decode_packed) the runtime's staged
// H2D copies do not overlap with the D2H copies of other streams (56.5 vs 55.0 ms), and at 1M records the extra
// streams / launches cost more than the overlap gains (7.1 vs 5.7 ms).  So: pinned source and >= 256 MB by default;
// RUHVRO_HIP_PIPELINE_MIN_MB overrides the threshold for both sources (tests force 0).
uint64_t pipeline_min_bytes(bool source_pinned) {      // read per call: tests switch it
  if (const char* e = std::getenv("RUHVRO_HIP_PIPELINE_MIN_MB")) return (uint64_t)std::strtoull(e, nullptr, 10) << 20;
  return source_pinned ? (256ull << 20) : ~0ull;
}

// One contiguous run of a call's chunks, decoded by one host thread on one device with its own stream and arenas.
struct Shard {
  uint32_t c0 = 0, c1 = 0;      // chunks [c0, c1) of the call
  int device = 0;
  uint32_t gate = 0;            // index of the turnstile pair of its device
  uint32_t ticket = 0;          // order among the shards of that device
};

int decode_host_impl(rh_schema* s, const Source& src, uint64_t n, uint64_t num_chunks, const rh_opts* opts,
                     ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats) {
  require_device();
  Timer total;
  Timeline::start();
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const uint32_t k = rh_clamp_chunks(n, num_chunks);
  std::memset(out_chunks, 0, sizeof(ArrowArray) * k);      // the failure paths release whatever was produced
  if (opts && opts->chunk_rows) throw std::invalid_argument("chunk_rows applies to rh_decode_device only");
  const bool multi = opts && opts->n_devices > 0;
  if (multi && !opts->devices) throw std::invalid_argument("n_devices > 0 with a NULL device list");
  int device = 0;
  if (!multi) {
    if (opts && opts->device >= 0) { HIPCHK(hipSetDevice(opts->device)); device = opts->device; }
    else HIPCHK(hipGetDevice(&device));
  }
  hipStream_t user_stream = opts ? (hipStream_t)opts->stream : nullptr;
  if (multi && user_stream) throw std::invalid_argument("a multi-device call runs on the engine's own streams (stream must be NULL)");
  // streaming hand-over (rh_opts.ready): the producer is still filling ptrs[] / lens[]; entries [0, *ready) are valid
  // (rh_opts.struct_size: a caller built against the ABI-3 struct, which ends before these two fields, leaves it 0)
  const bool has_handover = opts && opts->struct_size >= offsetof(rh_opts, gathered) + sizeof(uint64_t*);
  const uint64_t* const ready_ctr = (has_handover && src.slices()) ? opts->ready : nullptr;
  uint64_t* const gathered_ctr = (has_handover && src.slices()) ? opts->gathered : nullptr;
  auto wait_ready = [&](uint64_t upto) {
    if (!ready_ctr) return;
    for (uint32_t spins = 0;; spins++) {
      const uint64_t v = __atomic_load_n(ready_ctr, __ATOMIC_ACQUIRE);
      if (v == ~0ull) throw std::invalid_argument("the producer of the record slices gave up");
      if (v >= upto) return;
      if (spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  };
  const bool streaming = ready_ctr != nullptr && n >= 4096 && k >= 2 && !(opts && opts->stream);
  if (ready_ctr && !streaming) wait_ready(n);        // too small to pipeline: the classic call once everything is there
  uint64_t bytes = 0;
  if (streaming) {
    bytes = ~0ull >> 1;        // unknown yet: pipelined by construction (groups of chunks start as their entries arrive)
  } else if (src.slices()) {          // payload size decides whether the call is pipelined: a parallel sum of the lengths
    const unsigned nt = n >= (1u << 16) ? std::min(hw, 16u) : 1u;
    std::vector<uint64_t> part(nt, 0);
    run_threads(nt, [&](unsigned t) {
      uint64_t sum = 0;
      for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; i++) sum += src.lens[i];
      part[t] = sum;
    });
    for (uint64_t v : part) bytes += v;
  } else {
    bytes = n ? src.offsets[n] - src.offsets[0] : 0;
  }
  // slices are gathered into pinned memory shard by shard; so is a packed payload that is not pinned already
  bool packed_is_pinned = false;
  if (!src.slices() && src.data) {
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, src.data) == hipSuccess) packed_is_pinned = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();                // ordinary (unregistered) host memory: not an error here
  }
  const bool stage_packed = !src.slices() && !packed_is_pinned && env_long("RUHVRO_HIP_STAGE_PACKED", 1, 0, 1) != 0;
  const bool source_pinned = src.slices() || stage_packed || packed_is_pinned;

  // ---- the deal: which chunks go where
  std::vector<Shard> shards;
  std::vector<int> gate_device;          // one turnstile pair per distinct device
  if (multi) {
    // SURVEY 8(e) / rh_opts.devices: shard j of g gets chunks [j*k/g, (j+1)*k/g) on devices[j].  The shards of ONE device
    // share its PCIe link, so they pass its two copy directions in order (as the pipelined groups below do); shards of
    // different devices never wait for each other.
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    for (uint32_t j = 0; j < opts->n_devices; j++) {
      const int d = opts->devices[j];
      if (d < 0 || d >= ndev) throw std::invalid_argument("device ordinal " + std::to_string(d) + " out of range (" + std::to_string(ndev) + " visible)");
      Shard sh;
      sh.device = d;
      rh_shard_chunks(n, num_chunks, opts->n_devices, j, &sh.c0, &sh.c1, nullptr, nullptr);
      size_t gi = std::find(gate_device.begin(), gate_device.end(), d) - gate_device.begin();
      if (gi == gate_device.size()) gate_device.push_back(d);
      sh.gate = (uint32_t)gi;
      shards.push_back(sh);
    }
    std::vector<uint32_t> next_ticket(gate_device.size(), 0);
    for (Shard& sh : shards) sh.ticket = next_ticket[sh.gate]++;
  } else {
    // Large calls on the default stream are pipelined: chunks are independent (deserialize.rs:92-120), so contiguous
    // groups of chunks go through H2D -> kernels -> D2H on their own streams, staggered so that the link carries one
    // group's results out while the next group's records come in.
    const uint32_t groups = (user_stream == nullptr && k >= 2 && bytes >= pipeline_min_bytes(source_pinned)) ? std::min<uint32_t>(k, 8) : 1;
    if (groups <= 1) {
      decode_range(s, src, 0, n, num_chunks, nullptr, opts, device, user_stream, out_chunks, out_k, stats, nullptr, nullptr, 0,
                   bytes >= (4u << 20) ? std::min(hw, 16u) : 1u);
      if (stats) stats->total_ms = total.ms();
      return RH_OK;
    }
    gate_device.push_back(device);
    for (uint32_t g = 0; g < groups; g++) {
      Shard sh;
      sh.device = device;
      sh.c0 = (uint32_t)((uint64_t)k * g / groups);
      sh.c1 = (uint32_t)((uint64_t)k * (g + 1) / groups);
      sh.ticket = g;
      shards.push_back(sh);
    }
  }

  const uint64_t sz = n / k, rows_last = n - (uint64_t)(k - 1) * sz;
  const size_t ns = shards.size();
  // Record slices are gathered shard after shard by ONE pool of host threads, so the first shard is on the wire after
  // 1/ns of the gather time (side by side every shard would finish its gather at about the same, late, moment); each
  // shard's own thread waits for its block and takes it through H2D -> kernels -> D2H.
  const unsigned pack_threads = std::max(1u, std::min(hw, 32u));
  struct Ready {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> state;                 // 0 pending, 1 gathered, 2 failed
    std::vector<Gathered> block;
    std::vector<std::exception_ptr> err;
  } ready;
  ready.state.assign(ns, 0);
  ready.block.resize(ns);
  ready.err.resize(ns);
  std::thread gatherer;
  const bool pregather = src.slices() || stage_packed;
  if (pregather) {
    gatherer = std::thread([&] {
      struct PoolLoan {
        std::unique_ptr<CallPool> p;
        explicit PoolLoan(unsigned w) : p(HostCallCache::get().take_pool(w)) {}
        ~PoolLoan() { HostCallCache::get().give_pool(std::move(p)); }
      } loan(pack_threads);
      CallPool& pool = *loan.p;
      Timeline::mark(0, "gather pool up");
      auto par = [&](unsigned nt, const std::function<void(unsigned)>& f) { pool.parallel_for(nt, f); };
      for (size_t g = 0; g < ns; g++) {
        const Shard& sh = shards[g];
        char st = 1;
        try {
          if (sh.c1 > sh.c0) {
            HIPCHK(hipSetDevice(sh.device));
            const uint64_t r0 = (uint64_t)sh.c0 * sz, r1 = sh.c1 == k ? n : (uint64_t)sh.c1 * sz;
            if (streaming) { wait_ready(r1); Timeline::mark((uint32_t)g, "entries ready"); }
            ready.block[g] = src.slices() ? gather_slices(src, r0, r1 - r0, sh.device, pack_threads, par)
                                          : stage_packed_range(src, r0, r1 - r0, sh.device, pack_threads, par);
            if (gathered_ctr) __atomic_store_n(gathered_ctr, r1, __ATOMIC_RELEASE);   // (shards are gathered in row order)
            Timeline::mark((uint32_t)g, "gathered");
          }
        } catch (...) {
          ready.err[g] = std::current_exception();
          st = 2;
        }
        { std::lock_guard<std::mutex> l(ready.mu); ready.state[g] = st; }
        ready.cv.notify_all();
      }
    });
  }
  std::vector<rh_stats> gstats(ns);
  for (auto& gs : gstats) std::memset(&gs, 0, sizeof gs);
  std::vector<std::exception_ptr> failed(ns);
  std::vector<Turnstile> h2d_gates(gate_device.size()), d2h_gates(gate_device.size());
  const bool want = stats || (multi && opts->device_stats);
  rh_opts sopts = default_opts();
  sopts.flags = opts ? (opts->flags & 3) : 0;
  std::vector<std::thread> th;
  for (size_t g = 0; g < ns; g++) {
    th.emplace_back([&, g] {
      const Shard& sh = shards[g];
      hipStream_t st = nullptr;
      try {
        if (sh.c1 > sh.c0) {             // an empty shard (k < g) only passes its gates
          HIPCHK(hipSetDevice(sh.device));
          st = HostCallCache::get().take_stream(sh.device);
          const uint64_t r0 = (uint64_t)sh.c0 * sz, r1 = sh.c1 == k ? n : (uint64_t)sh.c1 * sz;
          ChunkGeo geo;
          geo.k = sh.c1 - sh.c0;
          geo.sz = sz;
          geo.rows_last = sh.c1 == k ? rows_last : sz;
          geo.payload_bytes = 0;           // decode_range fills it in
          Gathered* pre = nullptr;
          if (pregather) {
            std::unique_lock<std::mutex> l(ready.mu);
            ready.cv.wait(l, [&] { return ready.state[g] != 0; });
            if (ready.state[g] == 2) std::rethrow_exception(ready.err[g]);
            pre = &ready.block[g];
          }
          Timer tsh;
          decode_range(s, src, r0, r1, 0, &geo, &sopts, sh.device, st, out_chunks + sh.c0, nullptr,
                       want ? &gstats[g] : nullptr, &h2d_gates[sh.gate], &d2h_gates[sh.gate], sh.ticket, pack_threads, pre);
          gstats[g].total_ms = tsh.ms();
        }
      } catch (...) {
        failed[g] = std::current_exception();
      }
      h2d_gates[sh.gate].finish(sh.ticket);
      d2h_gates[sh.gate].finish(sh.ticket);
      if (st) HostCallCache::get().give_stream(sh.device, st, !failed[g]);      // (a shard that succeeded has waited for its stream)
    });
  }
  for (auto& t : th) t.join();
  if (gatherer.joinable()) gatherer.join();
  for (size_t g = 0; g < ns; g++) {
    if (!failed[g]) continue;
    for (uint32_t c = 0; c < k; c++)        // the call fails as a whole: drop what the other shards produced
      if (out_chunks[c].release) out_chunks[c].release(&out_chunks[c]);
    std::rethrow_exception(failed[g]);      // lowest shard = lowest rows: the error the serial order meets first
  }
  if (out_k) *out_k = k;
  if (multi && opts->device_stats)
    for (size_t g = 0; g < ns; g++) opts->device_stats[g] = gstats[g];
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    // stage times: shards of one device run one after the other through each stage (sum); devices run side by side (max)
    std::vector<rh_stats> per_dev(gate_device.size());
    for (auto& d : per_dev) std::memset(&d, 0, sizeof d);
    for (size_t g = 0; g < ns; g++) {
      const rh_stats& gs = gstats[g];
      stats->records += gs.records; stats->input_bytes += gs.input_bytes; stats->output_bytes += gs.output_bytes;
      stats->blocks += gs.blocks;
      rh_stats& d = per_dev[shards[g].gate];
      d.h2d_ms += gs.h2d_ms; d.size_kernel_ms += gs.size_kernel_ms; d.scan_kernel_ms += gs.scan_kernel_ms;
      stats->pack_ms += gs.pack_ms;                                // the shards are gathered one after the other
      d.emit_kernel_ms += gs.emit_kernel_ms; d.d2h_ms += gs.d2h_ms;
      if (gs.records) { stats->specialized = gs.specialized; stats->lds_bytes = gs.lds_bytes; }
    }
    for (const rh_stats& d : per_dev) {
      stats->h2d_ms = std::max(stats->h2d_ms, d.h2d_ms); stats->size_kernel_ms = std::max(stats->size_kernel_ms, d.size_kernel_ms);
      stats->scan_kernel_ms = std::max(stats->scan_kernel_ms, d.scan_kernel_ms);
      stats->emit_kernel_ms = std::max(stats->emit_kernel_ms, d.emit_kernel_ms); stats->d2h_ms = std::max(stats->d2h_ms, d.d2h_ms);
    }
    stats->chunks = k;
    stats->total_ms = total.ms();
  }
  return RH_OK;
}

}  // namespace

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
    if (pinned) ok = ok && hipHostMalloc((void**)&d.p, d.bytes, hipHostMallocDefault) == hipSuccess;
    else ok = ok && posix_memalign((void**)&d.p, 4096, d.bytes) == 0;
    if (ok) std::memset(d.p, 0, d.bytes);
  }
  double best = 1e30;
  for (uint32_t r = 0; ok && r < std::max(reps, 1u); r++) {
    Timer t;
    run_threads(shards, [&](unsigned j) {
      Dst& d = dst[j];
      gather_into(ptrs + d.rows0, lens + d.rows0, d.rows, threads_per_shard, d.p + 16, (uint64_t*)(d.p + d.o_off));
    });
    best = std::min(best, (double)t.ms());
  }
  for (Dst& d : dst)
    if (d.p) { if (pinned) (void)hipHostFree(d.p); else std::free(d.p); }
  *best_ms = best;
  return ok ? total : 0;
}

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
    *out = decode_device_impl(const_cast<rh_schema*>(s), (const uint8_t*)d_data, (const uint64_t*)d_offsets, data_len,
                              n, num_chunks, opts, stats);
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
    const unsigned parts = rh::kDecodeParts | (s->cs->encode_unsupported.empty() ? rh::kEncodeParts : 0u);
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
