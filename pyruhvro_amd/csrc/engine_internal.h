// Internals of the direct-decode engine, shared by its translation units:
//   engine.cpp              the C ABI (include/ruhvro_hip.h): schema objects, entry points, stats lines
//   engine_pools.cpp        device / pinned-host / control-block pools
//   engine_kernels.cpp      per-device schema programs, specialised kernels (code objects from kernel_jobs.h), error texts
//   engine_device_call.cpp  one device-resident decode call: the launch sequence, settle, the in-call split
//   engine_host.cpp         host in -> host out: gather, the pipelined chunk groups, the multi-GPU deal
//   engine_export.cpp       Arrow C Data / C Device Data export, host copies of results
//   engine_encode.cpp       Arrow -> Avro (rh_encode / rh_encode_device)
// Everything here lives in namespace rhe except the opaque C-ABI types (rh_schema, rh_device_result, rh_device_encoded).
#pragma once
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
int rh_launch_publish(void* ctrl, void* host, uint32_t head_words, uint32_t null_entries, uint32_t flag_word, uint32_t token, uint32_t nslots,
                      const uint32_t* tileflag, uint32_t nflags, uint32_t stat_word, void* stream);
int rh_launch_emit(const rh::KParams* P, uint32_t lds_bytes, void* stream, void* start, void* stop);
int rh_set_max_lds(uint32_t bytes);
uint32_t rh_lds_fixed_bytes(int K, int KL, int tile, int list_depth, int nnodes, int nbuf);
// Arrow -> Avro kernels (encode.hip)
int rh_launch_esize(const rh::EParams* P, uint32_t lds_bytes, void* stream);
int rh_launch_eemit(const rh::EParams* P, uint32_t lds_bytes, void* stream);
uint32_t rh_enc_lds_bytes(int ndom, int list_depth);
}
#include <memory>
namespace rhe {

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
struct NeedRanged {};      // a tile past the LDS window met specialised kernels without their ranged pair (LF_NEED_RANGED): repeat on the generic kernels
struct NeedTwoPass {};     // the single-pass form outgrew a column capacity (or needs what only the two-pass layout checks): repeat

extern std::atomic<uint64_t> g_counters[RH_CTR_COUNT];     // rh_engine_counters (include/ruhvro_hip.h)
constexpr uint32_t kRangedKeep = 64;      // calls a schema keeps launching its ranged kernels after the last tile past the LDS window
inline void count(int which, uint64_t by = 1) { g_counters[which].fetch_add(by, std::memory_order_relaxed); }

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw ::rhe::HipError(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr); \
  } while (0)

char* dup_msg(const std::string& s);
inline long env_long_early(const char* name, long dflt, long lo, long hi) {
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = std::strtol(e, &end, 10);
  return (end && *end == 0 && v >= lo && v <= hi) ? v : dflt;
}

// CPUs this process may use over time: hardware threads, cut down to its affinity mask and to its cgroup CPU quota
// (cpu.max / cfs_quota_us).  The GPU boxes report 256 hardware threads under a quota of 16 CPUs.  A quota bounds CPU TIME
// per period, not the width of a short burst: this sizes what runs for SECONDS (the kernel compile jobs), and is why
// idle threads here sleep instead of polling; the gather pool's bursts are sized by the hardware threads.
// RUHVRO_HIP_CPUS overrides.
unsigned effective_cpus();
const std::vector<std::vector<int>>& numa_node_cpus();      // cpus of every NUMA node (sysfs); empty when the kernel shows none
int device_numa_node(int device);                           // NUMA node of a HIP device, -1 unknown
void bind_thread_to_node(int node);                         // sched_setaffinity to that node's cpus (cached per thread)
// pinned blocks are pooled per NUMA node of the device they were allocated under (hipHostMalloc places them there)
inline int pin_key(int device) { const int n = device_numa_node(device); return n < 0 ? 0 : n; }

constexpr uint64_t kAlign = 256;
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// caching device / pinned-host pools (one per process; blocks are reused across calls so a
// steady-state decode does no hipMalloc)
// ---------------------------------------------------------------------------
// Pinned host blocks are written by kernels of ANY device of a multi-GPU call (control words, and the Arrow buffers themselves
// when a host call's results go straight into pinned memory): visible to every device, mapped into their address spaces.
constexpr unsigned kPinnedFlags = hipHostMallocPortable | hipHostMallocMapped;

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
    if (host_) device = pin_key(device);        // (pinned blocks: the NUMA node of the device; allocated with that device current)
    if (fail_injected()) throw HipError(std::string("HIP allocation of ") + std::to_string(size) + " bytes failed: injected failure (RUHVRO_HIP_FAIL_ALLOC)");
    {
      std::lock_guard<std::mutex> g(mu_);
      int best = -1;
      for (size_t i = 0; i < free_.size(); i++) {
        if (free_[i].device == device && free_[i].size >= size && free_[i].size <= size * 2 + (1 << 20)) {
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
    hipError_t e = host_ ? hipHostMalloc(&b.p, size, kPinnedFlags) : hipMalloc(&b.p, size);
    if (e != hipSuccess) {
      trim(0);
      e = host_ ? hipHostMalloc(&b.p, size, kPinnedFlags) : hipMalloc(&b.p, size);
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
  // RUHVRO_HIP_FAIL_ALLOC=N (test hook, read per request): the N-th block request from now on (N = 1: the next one), of either
  // pool and whether or not a cached block would have served it, fails the way an exhausted device does -- what a call does with a failed hipMalloc half-way through its
  // launch sequence (error class, no leaked lease, the next call unharmed: tests/test_round5.py).
  static bool fail_injected() {
    const char* e = std::getenv("RUHVRO_HIP_FAIL_ALLOC");
    if (!e || !*e) return false;
    static std::atomic<long> seen{0};
    static std::atomic<long> armed{0};
    const long n = std::atol(e);
    if (n <= 0) return false;
    if (armed.exchange(n) != n) seen.store(0);          // the hook was (re)set: count from here
    return seen.fetch_add(1) + 1 == n;
  }
  // A cached block of a suitable size, or an empty Block: never allocates.
  Block try_get(uint64_t size, int device) {
    size = align_up(std::max<uint64_t>(size, 1), 1 << 16);
    if (host_) device = pin_key(device);
    std::lock_guard<std::mutex> g(mu_);
    int best = -1;
    for (size_t i = 0; i < free_.size(); i++)
      if (free_[i].device == device && free_[i].size >= size && free_[i].size <= size * 2 + (1 << 20))
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
  // Pinned pool only: make sure a block of this size class is (or soon will be) cached -- allocated by a background thread,
  // because hipHostMalloc costs 0.2 ms per MB (tools/d2hcost.hip: 32 ms for a 1M-record call's results), never on a call's
  // own path.  No-op when a suitable block is cached, when `budget_left` cannot take it, or when enough requests are queued.
  void prefetch(uint64_t size, int device, uint64_t budget_left);
  uint64_t cached_bytes() { std::lock_guard<std::mutex> g(mu_); return cached_; }
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

Pool& dev_pool();
Pool& pin_pool();

struct Lease {   // RAII pool block
  Pool* pool = nullptr;
  Block b;
  Lease() = default;
  Lease(Pool& p, uint64_t size, int device) : pool(&p), b(p.get(size, device)) {}
  Lease(Pool& p, Block taken) : pool(&p), b(taken) {}       // a block the caller took from `p` itself (Pool::try_get)
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
CtrlPool& ctrl_pool();

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
  hipModule_t mod[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipFunction_t size_fn = nullptr, emit_fn = nullptr;
  // the ranged pair (tiles past the LDS window, spec_body.h ranged_tile): compiled and loaded when the schema first meets such
  // tiles; both or neither (emit_r_fn is stored last)
  std::atomic<hipFunction_t> size_r_fn{nullptr}, emit_r_fn{nullptr};
  std::atomic<bool> ranged_dead{false};
  // the single-pass form (decode kernels only): compiled and loaded when a call first asks for it, so it is written while
  // other calls of the schema read it
  std::atomic<hipFunction_t> fused_fn{nullptr};
  // (atomics: callers read these from the reference spec_kernel() returns, without the schema's mutex, while another thread of
  //  the same schema and device may be loading a kernel -- ADVICE round 5; `why` is only read once `dead` is seen set)
  std::atomic<bool> fused_dead{false};  // no such kernel for this schema (K > 64), or its compile failed
  std::atomic<bool> ok{false};          // size_fn and emit_fn are loaded
  std::atomic<bool> dead{false};        // they never will be: `why` says why (a failure is remembered)
  std::string why;
};

}  // namespace rhe

struct rh_schema {
  std::unique_ptr<rh::CompiledSchema> cs;
  std::mutex mu;
  std::map<int, rhe::DeviceProgram> dev;
  std::map<int, std::unique_ptr<rhe::SpecKernel>> spec;
  std::map<int, std::unique_ptr<rhe::SpecKernel>> espec;   // Arrow -> Avro kernels (rh_espec_size / rh_espec_emit)
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
  // Tiles past the LDS window: calls of this schema that still launch the ranged kernels behind the size / emit kernels -- set to
  // kRangedKeep by every settled call that met such tiles (rh_k_publish's tile statistics), counted down by the others.
  std::atomic<uint32_t> ranged_calls{0};
  uint32_t single_cooldown = 0, single_backoff = 0;      // calls the single pass sits out after a fail-over (8, 16, ... 1024; a success clears it)
};

namespace rhe {

const DeviceProgram& device_program(rh_schema* s, int device);
uint64_t spec_min_records();
// Specialised kernels of this schema on `device` (engine_kernels.cpp): never blocks on a compile unless the policy says so.
const SpecKernel& spec_kernel(rh_schema* s, int device, rh::CompilePolicy policy, bool encode = false, bool want_fused = false, bool want_ranged = false);
// What a call of `n` records may spend on kernels this schema does not have yet.
rh::CompilePolicy compile_policy(int mode, uint64_t n);
int launch_module(hipFunction_t f, const rh::KParams& P, uint32_t grid, uint32_t block, uint32_t lds, hipStream_t stream,
                  hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
std::string format_error(const rh::ErrInfo& e);

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

inline rh_opts default_opts() {
  rh_opts o;
  std::memset(&o, 0, sizeof o);
  o.device = -1;
  return o;
}

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  float ms() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace rhe

// ---------------------------------------------------------------------------
// device result
// ---------------------------------------------------------------------------
struct rh_decode_call;                 // one device-resident decode call (DeviceDecode below), still on its stream
struct rh_device_result {
  const rh::CompiledSchema* cs = nullptr;
  int device = 0;
  uint64_t n = 0, sz = 0, rows_last = 0;
  uint32_t k = 1;
  rhe::Lease arena;                         // all Arrow buffers of all chunks
  bool arena_host = false;             // the arena is pinned HOST memory the emit kernel wrote through the PCIe link (host calls): no D2H copy
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

struct rh_device_encoded {             // result of rh_encode_device: k BinaryArrays in HBM
  int device = 0;
  uint64_t n = 0, sz = 0, rows_last = 0;
  uint32_t k = 1;
  rhe::Lease out;                           // per chunk: i32 offsets[rows + 1] | data
  uint64_t out_bytes = 0;              // bytes of `out` in use
  std::vector<uint64_t> ooff;          // [k][2] offsets of the two buffers
  std::vector<uint64_t> data_bytes;    // [k] Avro bytes per chunk
  uint64_t exact = 0;
  uint64_t rows(uint32_t c) const { return n == 0 ? 0 : (c == k - 1 ? rows_last : sz); }
};

namespace rhe {

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

// Pinned host memory the engine may still lend to results: RUHVRO_HIP_PINNED_RESULT_MB (default 4096) less what live results
// hold -- what Pool::prefetch is allowed to add.  (Idle blocks are bounded separately, by the pool's own cache limit.)
inline uint64_t pinned_budget_left() {
  static const uint64_t bound = (uint64_t)env_long_early("RUHVRO_HIP_PINNED_RESULT_MB", 4096, 0, 1 << 20) << 20;
  const uint64_t used = Slab::pinned_result_bytes().load();
  return used >= bound ? 0 : bound - used;
}

void export_chunk(const rh_device_result& r, uint32_t c, const uint8_t* base, Slab* slab, ArrowArray* out);
void export_field(const rh::ArrowField& f, ArrowSchema* out);
// Device range -> freshly owned host memory (pooled pinned memory for large results)
Slab* slab_from_device(const uint8_t* dptr, uint64_t bytes, int device, hipStream_t stream = nullptr);
int to_host_impl(rh_device_result* r, ArrowArray* out_chunks, hipStream_t stream = nullptr);
// Settles an RH_ASYNC result (engine_device_call.cpp)
void settle(rh_device_result* r);

// ---------------------------------------------------------------------------
// the launch sequence
// ---------------------------------------------------------------------------
// integer knob from the environment, read at every use (tests change them inside one process); out of range = default
constexpr long kSinglePassDefault = 0;           // RUHVRO_HIP_SINGLE_PASS: 1 = every qualifying call prefers the single-pass form (else RH_SINGLE_PASS per call)
constexpr int kPublicFlags = 3 | RH_ASYNC | RH_TWO_PASS | RH_SINGLE_PASS;      // the rh_opts.flags bits of include/ruhvro_hip.h
constexpr int RH_INTERNAL_HOST_ARENA = 0x200;    // rh_opts.flags, engine-internal (host calls): write the Arrow buffers straight into pinned host memory if a pooled block is free
constexpr int RH_INTERNAL_TWO_PASS = 0x100;      // rh_opts.flags, engine-internal: this call must take the two-pass path
constexpr long kInternalStreamsDefault = 1;      // RUHVRO_HIP_INTERNAL_STREAMS (decode_device_split)
constexpr long kSplitMinDefault = 1000000;       // RUHVRO_HIP_SPLIT_MIN: records below which a call is never split

inline long env_long(const char* name, long dflt, long lo, long hi) {
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
                                     const ChunkGeo* geo = nullptr);

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

void require_device();

// Where a call's records are: packed (one payload + n+1 absolute offsets: rh_decode_packed, what the reference builds
// at deserialize.rs:90) or one (pointer, length) slice per record (rh_decode: what src/lib.rs:29-33 extracts).
struct Source {
  const uint8_t* data = nullptr;
  const uint64_t* offsets = nullptr;
  const uint8_t* const* ptrs = nullptr;
  const uint64_t* lens = nullptr;
  bool slices() const { return ptrs != nullptr || (data == nullptr && offsets == nullptr); }
};

void run_threads(unsigned nt, const std::function<void(unsigned)>& f);
int decode_host_impl(rh_schema* s, const Source& src, uint64_t n, uint64_t num_chunks, const rh_opts* opts,
                     ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats);
uint64_t gather_into(const uint8_t* const* ptrs, const uint64_t* lens, uint64_t n, unsigned nt_in, uint8_t* hdst, uint64_t* hoff);

}  // namespace rhe
