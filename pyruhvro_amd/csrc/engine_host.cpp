// Host in -> host out (rh_decode / rh_decode_packed): gather of the record slices into pinned memory, the pipelined chunk
// groups of one device, the multi-GPU deal (rh_opts.devices).  Replaces the reference's chunk driver
// (ruhvro/src/deserialize.rs:76-121: pack, slice, one task per chunk, ordered join).
#include "engine_internal.h"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace rhe {

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

void run_threads(unsigned nt, const std::function<void(unsigned)>& f) {
  if (nt <= 1) { f(0); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back(f, t);
  for (auto& x : th) x.join();
}

// A pool of host threads: parallel_for hands out task indices to the workers (and to the caller) and returns when all are
// done.  The gather of a pipelined call runs group after group on it, two short phases per group, so what a phase costs
// beyond its work decides the call: with a mutex + condition variable per phase a 16 MB group took 0.6 ms to gather on 32
// threads (RUHVRO_HIP_TIMELINE, profiles/r05c_*: the gather chain, not the PCIe link, bounded a 1M-record call).  Here the
// hand-over is lock-free -- one 64-bit ticket word carries the phase's generation in its high half and the next task index
// in its low half; a claim is one fetch_add -- and workers that run out of tasks poll the ticket for ~50 us (the second
// phase of a group follows the first at once) before they sleep on a futex, from which all of them wake side by side.
class CallPool {
 public:
  explicit CallPool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) th_.emplace_back([this] { work(); });
  }
  ~CallPool() {
    stop_.store(true, std::memory_order_seq_cst);
    gen32_.fetch_add(0x80000000u, std::memory_order_seq_cst);      // (any change of the word ends a futex wait)
    futex_wake_all();
    for (auto& t : th_) t.join();
  }
  unsigned workers() const { return (unsigned)th_.size(); }
  void parallel_for(unsigned ntasks, const std::function<void(unsigned)>& f) {     // one caller at a time
    if (ntasks == 0) return;
    const uint64_t g = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
    Slot& sl = slot_[g & 1];
    sl.fn.store(&f, std::memory_order_relaxed);
    sl.ntasks.store(ntasks, std::memory_order_relaxed);
    left_.store(ntasks, std::memory_order_relaxed);
    ticket_.store(g << 32, std::memory_order_seq_cst);
    gen32_.store((uint32_t)g, std::memory_order_seq_cst);
    if (sleepers_.load(std::memory_order_seq_cst) > 0) futex_wake_all();
    run_tasks();                                     // the caller takes tasks too
    for (unsigned spins = 0; left_.load(std::memory_order_acquire) != 0; spins++) {
      if (spins < 4096) cpu_relax();
      else std::this_thread::yield();
    }
  }

 private:
  struct Slot { std::atomic<const std::function<void(unsigned)>*> fn{nullptr}; std::atomic<unsigned> ntasks{0}; };
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  // Claims tasks of the current phase.  A claim is interpreted by the generation bits IT returned, so a thread that is late
  // by a phase takes a task of the phase it finds, never one of a phase that is over.
  void run_tasks() {
    for (;;) {
      const uint64_t t = ticket_.fetch_add(1, std::memory_order_acq_rel);
      const uint64_t tg = t >> 32;
      const unsigned idx = (unsigned)(t & 0xFFFFFFFFu);
      if (tg == 0) return;                                 // (no phase yet)
      const Slot& sl = slot_[tg & 1];                      // (rewritten only by generation tg + 2: ruled out below)
      const unsigned ntasks = sl.ntasks.load(std::memory_order_relaxed);
      const std::function<void(unsigned)>* fn = sl.fn.load(std::memory_order_relaxed);
      // phase tg ended while this thread looked: an unexecuted task keeps its phase open, so idx was beyond its tasks
      if ((ticket_.load(std::memory_order_acquire) >> 32) != tg) continue;
      if (idx >= ntasks) return;
      (*fn)(idx);
      left_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void work() {
    uint64_t seen = 0;
    for (;;) {
      const uint64_t g = ticket_.load(std::memory_order_acquire) >> 32;
      if (g != seen) {
        seen = g;
        run_tasks();
        continue;
      }
      // nothing new: poll for ~50 us (the next phase of the same group follows at once), then sleep on the generation word.
      // (Polling through the gaps BETWEEN groups was measured and thrown out: 32 spinning threads slowed the caller's own
      //  extraction threads on the GPU box -- a 1M-record call went from 6.7 to 9.2 ms, profiles/r05d_*.)
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      for (unsigned spins = 0;; spins++) {
        if (stop_.load(std::memory_order_relaxed)) return;
        if ((ticket_.load(std::memory_order_acquire) >> 32) != seen) { got = true; break; }
        cpu_relax();
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
      }
      if (got) continue;
      sleepers_.fetch_add(1, std::memory_order_seq_cst);
      const uint32_t word = gen32_.load(std::memory_order_seq_cst);
      if (word == (uint32_t)seen && !stop_.load(std::memory_order_seq_cst)) futex_wait(word);      // (returns at once if the word moved on)
      sleepers_.fetch_sub(1, std::memory_order_seq_cst);
      if (stop_.load(std::memory_order_relaxed)) return;
    }
  }
  // Sleep / wake on gen32_ without a mutex: the woken workers do not queue up behind one another to re-acquire a lock
  // (condition_variable wake-ups of 32 waiters are serialised by the mutex they all return through).
  void futex_wait(uint32_t expected) {
    ::syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen32_), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
  }
  void futex_wake_all() {
    ::syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen32_), FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
  }
  std::vector<std::thread> th_;
  std::atomic<uint64_t> ticket_{0};
  Slot slot_[2];
  std::atomic<unsigned> left_{0};
  std::atomic<unsigned> sleepers_{0};
  std::atomic<bool> stop_{false};
  std::atomic<uint32_t> gen32_{0};        // low half of the generation: the futex word
  static_assert(sizeof(std::atomic<uint32_t>) == 4, "futex word");
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
  // the pool's threads work next to the staging block: hipHostMalloc puts it on the NUMA node of the shard's GPU, the tasks bind
  // their thread there (bind_thread_to_node: one syscall when a thread changes node, none otherwise) -- on the two-socket GPU box
  // g shards gathering at once reach 148-156 GB/s placed against 98-106 unplaced (profiles/r05n_gather_scaling.json)
  const int node = nt > 1 ? device_numa_node(device) : -1;
  std::vector<uint64_t> part(nt + 1, 0);
  par(nt, [&](unsigned t) {
    bind_thread_to_node(node);
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
    bind_thread_to_node(node);
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
  const int node = nt > 1 ? device_numa_node(device) : -1;
  par(nt, [&](unsigned t) {
    bind_thread_to_node(node);
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
  // kernel form only (the host paths settle every device call themselves) + the results may be written straight into pinned
  // host memory (rh_decode_call::arena_lease)
  o.flags = (opts ? (opts->flags & 3) : 0) | RH_INTERNAL_HOST_ARENA;
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
  if (r->arena_host) {             // nothing to copy: the result's buffers are in host memory already
    Timer td;
    to_host_impl(r.get(), out_chunks, stream);
    d2h = td.ms();
    Timeline::mark(ticket, "adopted");
  } else {
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

// Payload bytes from which a call is pipelined (contiguous groups of chunks through gather -> H2D -> kernels on their own
// streams).  Round 5, MI355X box (profiles/r05p_pipeline_threshold.txt): with the results written straight into pinned host
// memory the pipeline pays from a few tens of MB -- record slices (gathered into pinned memory anyway): 32 MB 3.06 -> 2.1 ms,
// 128 MB (1M benchmark records) 8.4 -> 5.4 ms; a packed payload in pageable memory (staged through pinned memory by the
// call's threads when pipelined, one runtime-staged copy when not): 32 MB 1.52 -> 1.72 ms, 128 MB 5.67 -> 5.0 ms.
// (Round 1 measured the opposite at 1M records, 7.1 vs 5.7 ms: the D2H copies of one group and the H2D copies of the next
// then shared one copy queue and ran strictly one after the other, profiles/r05o_host_pipeline_trace_d2h_copies.txt.)
// RUHVRO_HIP_PIPELINE_MIN_MB overrides the threshold for both sources (tests force 0).
uint64_t pipeline_min_bytes(bool source_pinned, bool staged_pageable) {      // read per call: tests switch it
  if (const char* e = std::getenv("RUHVRO_HIP_PIPELINE_MIN_MB")) return (uint64_t)std::strtoull(e, nullptr, 10) << 20;
  if (staged_pageable) return 96ull << 20;
  return source_pinned ? (24ull << 20) : ~0ull;
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
    const uint32_t groups = (user_stream == nullptr && k >= 2 && bytes >= pipeline_min_bytes(source_pinned, stage_packed)) ? std::min<uint32_t>(k, 8) : 1;
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
  // (sized by the hardware threads, not by effective_cpus(): a cgroup quota bounds CPU TIME per period, not how many cores a
  //  short burst may use -- the gather of a group is such a burst, and on the GPU box (256 threads, quota 16) 32 gather threads
  //  beat 12: record_slices 40.6 vs 45.7 ms per 10M records, profiles/r05g_*.  RUHVRO_HIP_GATHER_THREADS: A/B knob.)
  const unsigned pack_threads = (unsigned)env_long("RUHVRO_HIP_GATHER_THREADS", (long)std::max(1u, std::min(std::max(1u, std::thread::hardware_concurrency()), 32u)), 1, 256);
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


}  // namespace rhe

// Test hook (include/ruhvro_hip.h): `phases` parallel_for phases of 1..max_tasks tasks on a CallPool of `workers` threads,
// with pauses between some of them so that workers go through poll -> sleep -> wake; every task index of every phase must
// run exactly once.  0 = ok, else the 1-based phase that went wrong.
extern "C" uint32_t rh_selftest_pool(uint32_t workers, uint32_t phases, uint32_t max_tasks) {
  using namespace rhe;
  CallPool pool(std::max(1u, workers));
  std::vector<std::atomic<uint32_t>> hits(std::max(1u, max_tasks));
  uint64_t x = 88172645463325252ull;
  for (uint32_t ph = 1; ph <= phases; ph++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const unsigned nt = 1 + (unsigned)(x % std::max(1u, max_tasks));
    for (auto& h : hits) h.store(0, std::memory_order_relaxed);
    std::atomic<uint64_t> sum{0};
    pool.parallel_for(nt, [&](unsigned t) {
      hits[t].fetch_add(1, std::memory_order_relaxed);
      sum.fetch_add(t + 1, std::memory_order_relaxed);
    });
    for (unsigned t = 0; t < nt; t++)
      if (hits[t].load(std::memory_order_relaxed) != 1) return ph;
    for (unsigned t = nt; t < max_tasks; t++)
      if (hits[t].load(std::memory_order_relaxed) != 0) return ph;
    if (sum.load() != (uint64_t)nt * (nt + 1) / 2) return ph;
    if ((x >> 20) % 64 == 0) std::this_thread::sleep_for(std::chrono::microseconds(1500));     // let the workers fall asleep
  }
  return 0;
}
