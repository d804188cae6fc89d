// CPython extension `_pyruhvro`: the GIL-side half of the Python boundary.
//
// Mirrors what the reference's PyO3 layer does around the native call
// (src/lib.rs:29-33 extract_bytes_list, 64-68/82-86 py.detach, 25-27 error
// mapping): borrow (ptr, len) of every `bytes` in the list while holding the
// GIL and a strong reference, release the GIL, call the C ABI
// (include/ruhvro_hip.h), and hand the resulting Arrow C structs to Python as
// raw addresses that pyarrow imports (`RecordBatch._import_from_c`).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <thread>
#include <system_error>
#include <atomic>
#include <string>
#include <cstddef>
#include <memory>
#include <vector>

#include "ruhvro_hip.h"

namespace {

const char* kCapsule = "ruhvro_hip.schema";

void capsule_free(PyObject* cap) {
  rh_schema* s = (rh_schema*)PyCapsule_GetPointer(cap, kCapsule);
  if (s) rh_schema_free(s);
}

PyObject* raise_from(int rc, char* err) {
  std::string msg = err ? err : "ruhvro_hip error";
  if (err) rh_free_string(err);
  PyObject* exc = PyExc_RuntimeError;
  if (rc == RH_ERR_SCHEMA || rc == RH_ERR_DECODE) exc = PyExc_ValueError;   // src/lib.rs:25-27
  else if (rc == RH_ERR_ARGUMENT) exc = PyExc_ValueError;
  PyErr_SetString(exc, msg.c_str());
  return nullptr;
}

PyObject* py_compile_schema(PyObject*, PyObject* args) {
  const char* s;
  Py_ssize_t n;
  if (!PyArg_ParseTuple(args, "s#", &s, &n)) return nullptr;
  char* err = nullptr;
  rh_schema* h = rh_schema_compile(s, (size_t)n, &err);
  if (!h) return raise_from(RH_ERR_SCHEMA, err);
  return PyCapsule_New(h, kCapsule, capsule_free);
}

rh_schema* get_schema(PyObject* cap) {
  return (rh_schema*)PyCapsule_GetPointer(cap, kCapsule);
}

PyObject* py_schema_ptr(PyObject*, PyObject* args) {
  PyObject* cap;
  if (!PyArg_ParseTuple(args, "O", &cap)) return nullptr;
  rh_schema* s = get_schema(cap);
  if (!s) return nullptr;
  return PyLong_FromVoidPtr(s);
}

// export_schema(capsule) -> address of a malloc'd ArrowSchema (import with pyarrow, then free_struct)
PyObject* py_export_schema(PyObject*, PyObject* args) {
  PyObject* cap;
  if (!PyArg_ParseTuple(args, "O", &cap)) return nullptr;
  rh_schema* s = get_schema(cap);
  if (!s) return nullptr;
  ArrowSchema* out = (ArrowSchema*)std::calloc(1, sizeof(ArrowSchema));
  if (rh_schema_export(s, out) != RH_OK) {
    std::free(out);
    PyErr_SetString(PyExc_RuntimeError, "schema export failed");
    return nullptr;
  }
  return PyLong_FromVoidPtr(out);
}

PyObject* py_free_struct(PyObject*, PyObject* args) {
  PyObject* addr;
  if (!PyArg_ParseTuple(args, "O", &addr)) return nullptr;
  void* p = PyLong_AsVoidPtr(addr);
  if (!p && PyErr_Occurred()) return nullptr;
  std::free(p);
  Py_RETURN_NONE;
}

PyObject* stats_dict(const rh_stats& st) {
  return Py_BuildValue("{s:K,s:K,s:K,s:I,s:I,s:f,s:f,s:f,s:f,s:f,s:f,s:f,s:I,s:I}", "records",
                       (unsigned long long)st.records, "input_bytes", (unsigned long long)st.input_bytes,
                       "output_bytes", (unsigned long long)st.output_bytes, "chunks", st.chunks, "blocks", st.blocks,
                       "pack_ms", st.pack_ms, "h2d_ms", st.h2d_ms, "size_kernel_ms", st.size_kernel_ms,
                       "scan_kernel_ms", st.scan_kernel_ms, "emit_kernel_ms", st.emit_kernel_ms, "d2h_ms", st.d2h_ms,
                       "total_ms", st.total_ms, "specialized", st.specialized, "lds_bytes", st.lds_bytes);
}

// A buffer of n 8-byte words from a cache of at most four idle ones (taken and returned with the GIL held, so no lock);
// blocks beyond PYRUHVRO_SCRATCH_MAX_MB (default 512) are not kept.
class Scratch {
 public:
  explicit Scratch(size_t words) : words_(words) {
#ifdef Py_GIL_DISABLED
    cap_ = words; p_ = std::malloc(cap_ * 8); return;      // free-threaded CPython: no GIL guards the cache, so there is none
#endif
    auto& idle = cache();
    size_t best = idle.size();
    for (size_t i = 0; i < idle.size(); i++)
      if (idle[i].second >= words && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
    if (best < idle.size()) {
      p_ = idle[best].first; cap_ = idle[best].second;
      idle.erase(idle.begin() + (long)best);
    } else {
      cap_ = words;
      p_ = std::malloc(cap_ * 8);
    }
  }
  bool ok() const { return p_ != nullptr; }
  ~Scratch() {
    static const size_t max_words = [] {
      const char* e = std::getenv("PYRUHVRO_SCRATCH_MAX_MB");
      return (size_t)(e && *e ? std::atol(e) : 512l) * (1u << 20) / 8;
    }();
#ifdef Py_GIL_DISABLED
    std::free(p_); return;
#endif
    auto& idle = cache();
    if (p_ && cap_ <= max_words) {
      if (idle.size() >= 4) {                 // full: the smallest block makes room (a process that went from small calls to
        size_t small = 0;                     // large ones must end up caching the large blocks -- they are the expensive ones)
        for (size_t i = 1; i < idle.size(); i++)
          if (idle[i].second < idle[small].second) small = i;
        if (idle[small].second >= cap_) { std::free(p_); return; }
        std::free(idle[small].first);
        idle.erase(idle.begin() + (long)small);
      }
      idle.emplace_back(p_, cap_);
    } else {
      std::free(p_);
    }
  }
  Scratch(const Scratch&) = delete;
  Scratch& operator=(const Scratch&) = delete;
  template <typename T> T* as() const { static_assert(sizeof(T) == 8, "8-byte words"); return (T*)p_; }
  template <typename T> T& at(size_t i) const { return ((T*)p_)[i]; }

 private:
  static std::vector<std::pair<void*, size_t>>& cache() {
    static auto* v = new std::vector<std::pair<void*, size_t>>();
    return *v;
  }
  void* p_ = nullptr;
  size_t words_ = 0, cap_ = 0;
};

// Phase split of the calling thread's most recent decode() (last_decode_profile(); PYRUHVRO_PYPROF=1 prints the same line):
// milliseconds of set-up, of the list extraction, of the rest of the call, and of the call's time with the GIL held.
struct DecodeProfile { double n = 0, streaming = 0, alloc = 0, extract = 0, tail = 0, total = 0, gil_held = 0; };
thread_local DecodeProfile g_last_profile;

// decode(capsule, list, num_chunks, device=-1, stream=0, want_stats=False, kernel=0, devices=None)
//   devices: None, or a sequence of HIP device ordinals to shard the chunks over (rh_opts.devices)
//   -> (list[int] addresses of malloc'd ArrowArray structs, stats dict | None)
PyObject* py_decode(PyObject*, PyObject* args) {
  PyObject *cap, *list;
  unsigned long long num_chunks;
  int device = -1;
  unsigned long long stream = 0;
  int want_stats = 0;
  int kernel = RH_KERNEL_AUTO;
  PyObject* devs = Py_None;
  if (!PyArg_ParseTuple(args, "OOK|iKpiO", &cap, &list, &num_chunks, &device, &stream, &want_stats, &kernel, &devs)) return nullptr;
  rh_schema* s = get_schema(cap);
  if (!s) return nullptr;
  std::vector<int32_t> devices;
  if (devs != Py_None) {
    PyObject* seq = PySequence_Fast(devs, "argument 'devices': expected a sequence of ints");
    if (!seq) return nullptr;
    for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); i++) {
      const long d = PyLong_AsLong(PySequence_Fast_GET_ITEM(seq, i));
      if (d == -1 && PyErr_Occurred()) { Py_DECREF(seq); return nullptr; }
      devices.push_back((int32_t)d);
    }
    Py_DECREF(seq);
  }
  if (!PyList_Check(list)) {
    PyErr_SetString(PyExc_TypeError, "argument 'list': expected a list of bytes");
    return nullptr;
  }
  const Py_ssize_t n = PyList_GET_SIZE(list);
  // PYRUHVRO_PYPROF=1: one stderr line per call with the milliseconds of the boundary's own phases
  static const bool pyprof = [] { const char* e = std::getenv("PYRUHVRO_PYPROF"); return e && *e && *e != '0'; }();
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  double ms_alloc = 0, ms_extract = 0, ms_tail = 0, ms_released = 0;
  (void)ms_alloc;
  // one (pointer, length) per record, uninitialised (2 x 8 bytes x n): every slot is written below.  A reference is
  // held on every bytes object while the GIL is released; the object is recovered from its payload pointer afterwards
  // (payload = object + offsetof(PyBytesObject, ob_sval)), so no third array is kept.
  // (the two arrays come from a small cache kept between calls, guarded by the GIL: fresh ones cost a page fault per 4 KiB when
  //  they are filled and a munmap when they are dropped -- 160 MB per 10M-record call, 17 ms of a 60 ms call on the GPU box,
  //  scripts/host_gap_probe.py)
  Scratch ptrs_s((size_t)n + 1), lens_s((size_t)n + 1);
  if (!ptrs_s.ok() || !lens_s.ok()) return PyErr_NoMemory();
  const uint8_t** const ptrs = ptrs_s.as<const uint8_t*>();
  uint64_t* const lens = lens_s.as<uint64_t>();
  constexpr size_t kPayload = offsetof(PyBytesObject, ob_sval);
  auto drop_range = [&](Py_ssize_t from, Py_ssize_t upto) {
    constexpr Py_ssize_t kAheadD = 24;
    for (Py_ssize_t i = from; i < upto; i++) {
      if (i + kAheadD < upto) __builtin_prefetch(ptrs[(size_t)(i + kAheadD)] - kPayload, 1, 1);
      PyObject* o = (PyObject*)(ptrs[(size_t)i] - kPayload);
      Py_DECREF(o);
    }
  };
  const uint32_t k = rh_clamp_chunks((uint64_t)n, num_chunks);
  ArrowArray* chunks = (ArrowArray*)std::calloc(k, sizeof(ArrowArray));
  rh_opts opts;
  std::memset(&opts, 0, sizeof opts);
  opts.struct_size = (uint32_t)sizeof opts;
  opts.device = device;
  opts.flags = kernel;
  opts.stream = (void*)(uintptr_t)stream;
  if (!devices.empty()) { opts.devices = devices.data(); opts.n_devices = (uint32_t)devices.size(); }
  rh_stats st;
  std::memset(&st, 0, sizeof st);
  char* err = nullptr;
  uint32_t out_k = 0;
  int rc = RH_OK;
  // Large lists are BORROWED and handed over while they are still being extracted (rh_opts.ready / gathered):
  //  * the engine runs on its own thread from the start and gathers a chunk group into pinned memory as soon as its
  //    (pointer, length) entries exist, so the copies and the kernels of the first groups overlap the extraction;
  //  * the entries are read by a few helper threads, READ-ONLY: this thread holds the GIL until the engine reports that
  //    it has copied every record's bytes (`gathered == n`), so no object of the list can change or die meanwhile and no
  //    reference is taken -- the two passes over 2M object headers that taking and dropping references cost were 2/3 of
  //    the serial call (scripts/pyprof_list_bytes.py); bytearray elements are borrowed the same way;
  //  * from then on nothing points into the Python heap any more and the GIL is released for the rest of the call
  //    (py.detach(...), src/lib.rs:82-86).  The GIL is held for a few milliseconds per million records -- less than the
  //    reference holds it for its own extraction (extract_bytes_list, src/lib.rs).
  // Small lists take the classic form: one reference per object, GIL released around the whole engine call.
  static const long stream_min = [] {
    const char* e = std::getenv("PYRUHVRO_STREAM_MIN");
    return e && *e ? std::atol(e) : 65536l;
  }();
  bool streaming = stream_min >= 0 && (long)n >= stream_min && n >= 2 && stream == 0;
#ifdef Py_GIL_DISABLED
  streaming = false;      // free-threaded CPython: holding "the GIL" protects nothing, so objects are never borrowed without a reference
#endif
  // "-1 = the current device" means the CALLER's current device (torch.cuda.set_device / hipSetDevice are per host
  // thread): the streaming form runs the engine on a thread of its own, where the current device would be 0 again, so
  // the ordinal is resolved here, on the calling thread, before that thread exists.
  if (streaming && opts.device < 0 && devices.empty()) {
    const int cur = rh_current_device();
    if (cur >= 0) opts.device = cur;
  }
  std::atomic<uint64_t> ready{0}, gathered{0};
  std::atomic<bool> finished{false};
  std::thread worker;
  Py_ssize_t done = 0;
  bool ok = true;
  ms_alloc = ms_since(t_start);
  const auto t_extract = std::chrono::steady_clock::now();
  std::chrono::steady_clock::time_point t_tail;
  if (streaming) {
    static_assert(sizeof(std::atomic<uint64_t>) == sizeof(uint64_t), "plain 64-bit atomics");
    opts.ready = reinterpret_cast<const uint64_t*>(&ready);
    opts.gathered = reinterpret_cast<uint64_t*>(&gathered);
    try {
      worker = std::thread([&] {                     // (never touches Python)
        rc = rh_decode(s, ptrs, lens, (uint64_t)n, num_chunks, &opts, chunks, &out_k, want_stats ? &st : nullptr, &err);
        finished.store(true, std::memory_order_release);
      });
    } catch (const std::system_error&) {             // no thread to be had: the classic form below
      streaming = false;
      opts.ready = nullptr;
      opts.gathered = nullptr;
    }
  }
  if (streaming) {
    // read-only extraction, blocks of kBlk entries claimed in order by the helpers; `ready` follows the done prefix
    constexpr Py_ssize_t kBlk = 16384;
    const Py_ssize_t nblk = (n + kBlk - 1) / kBlk;
    const unsigned nth = (unsigned)std::max<long>(1, std::min<long>({8l, (long)std::thread::hardware_concurrency() / 4, (long)nblk}));
    std::unique_ptr<std::atomic<unsigned char>[]> blk_done;
    try {
      blk_done.reset(new std::atomic<unsigned char>[(size_t)nblk]);
    } catch (const std::bad_alloc&) {                 // the engine thread is running: tell it to give up, join it, then report
      ready.store(~0ull, std::memory_order_release);
      Py_BEGIN_ALLOW_THREADS
      worker.join();
      Py_END_ALLOW_THREADS
      for (uint32_t c = 0; c < k; c++)
        if (chunks[c].release) chunks[c].release(&chunks[c]);
      if (err) rh_free_string(err);
      std::free(chunks);
      return PyErr_NoMemory();
    }
    for (Py_ssize_t b = 0; b < nblk; b++) blk_done[(size_t)b].store(0, std::memory_order_relaxed);
    std::atomic<Py_ssize_t> first_bad{n};
    PyObject** items = PySequence_Fast_ITEMS(list);          // (a list: its item array, stable while the GIL is held)
    std::atomic<Py_ssize_t> next_blk{0};
    // `ready` follows the prefix of finished blocks WHILE the blocks are extracted: whoever finishes a block advances the
    // prefix as far as it goes (round 3 published it only after the calling thread had run out of blocks to claim, i.e.
    // at the end of the extraction -- the engine's first gather then started when the last entry was there, and nothing
    // of the extraction overlapped the copies: 10M records 58.9 -> see profiles/r04j_python_surface.txt)
    std::atomic<Py_ssize_t> prefix{0};
    auto publish = [&]() {
      for (;;) {
        Py_ssize_t p = prefix.load(std::memory_order_acquire);
        if (p >= nblk || !blk_done[(size_t)p].load(std::memory_order_acquire)) return;
        if (!prefix.compare_exchange_strong(p, p + 1, std::memory_order_acq_rel)) continue;       // another thread moved it
        const Py_ssize_t bad = first_bad.load(std::memory_order_relaxed);
        const uint64_t upto = (uint64_t)std::min<Py_ssize_t>(std::min<Py_ssize_t>((p + 1) * kBlk, n), bad);
        uint64_t cur = ready.load(std::memory_order_relaxed);
        while (cur < upto && !ready.compare_exchange_weak(cur, upto, std::memory_order_release)) {}   // monotonic
      }
    };
    auto extract = [&]() {
      for (Py_ssize_t b; (b = next_blk.fetch_add(1, std::memory_order_relaxed)) < nblk;) {
        const Py_ssize_t lo = b * kBlk, hi = std::min<Py_ssize_t>(lo + kBlk, n);
        if (first_bad.load(std::memory_order_relaxed) < lo) break;           // a lower element is already known bad
        for (Py_ssize_t i = lo; i < hi; i++) {
          if (i + 24 < hi) __builtin_prefetch(items[i + 24], 0, 1);
          PyObject* it = items[i];
          if (PyBytes_Check(it)) {
            ptrs[(size_t)i] = (const uint8_t*)PyBytes_AS_STRING(it);
            lens[(size_t)i] = (uint64_t)PyBytes_GET_SIZE(it);
          } else if (PyByteArray_Check(it)) {
            ptrs[(size_t)i] = (const uint8_t*)PyByteArray_AS_STRING(it);
            lens[(size_t)i] = (uint64_t)PyByteArray_GET_SIZE(it);
          } else {
            Py_ssize_t cur = first_bad.load(std::memory_order_relaxed);
            while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
            break;
          }
        }
        blk_done[(size_t)b].store(1, std::memory_order_release);
        publish();
      }
    };
    std::vector<std::thread> helpers;
    try {
      for (unsigned t = 1; t < nth; t++) helpers.emplace_back(extract);
    } catch (const std::system_error&) {}            // fewer helpers: the blocks are claimed dynamically
    // this thread takes its share too, then follows the prefix of finished blocks
    extract();
    for (;;) {                                        // (the helpers' last blocks)
      publish();
      if (prefix.load(std::memory_order_acquire) == nblk || first_bad.load(std::memory_order_relaxed) < n) break;
      std::this_thread::yield();
    }
    for (auto& h : helpers) h.join();
    const Py_ssize_t bad = first_bad.load();
    ms_extract = ms_since(t_extract);
    t_tail = std::chrono::steady_clock::now();
    if (bad < n) {
      ready.store(~0ull, std::memory_order_release);          // the engine call fails; nothing of it is used
      Py_BEGIN_ALLOW_THREADS
      worker.join();
      Py_END_ALLOW_THREADS
      for (uint32_t c = 0; c < k; c++)
        if (chunks[c].release) chunks[c].release(&chunks[c]);
      if (err) rh_free_string(err);
      std::free(chunks);
      PyErr_Format(PyExc_TypeError, "list element %zd: expected bytes, got %s", bad, Py_TYPE(items[bad])->tp_name);
      return nullptr;
    }
    ready.store((uint64_t)n, std::memory_order_release);
    // the GIL stays with this thread until the engine has copied every record (or gave up): then nothing borrowed is in use
    while (!finished.load(std::memory_order_acquire) && gathered.load(std::memory_order_acquire) < (uint64_t)n)
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    const auto t_rel = std::chrono::steady_clock::now();
    Py_BEGIN_ALLOW_THREADS
    worker.join();
    Py_END_ALLOW_THREADS
    ms_released = ms_since(t_rel);
    done = n;
  } else {
    // The list's objects are scattered over the heap: one cache miss per header.  The pointer array is contiguous, so
    // the headers 24 items ahead are prefetched while this one is read (a shuffled 2M-record list: 134 -> 103 ms).
    constexpr Py_ssize_t kAhead = 24;
    for (Py_ssize_t i = 0; i < n; i++) {
      if (i + kAhead < n) __builtin_prefetch(PyList_GET_ITEM(list, i + kAhead), 1, 1);
      PyObject* it = PyList_GET_ITEM(list, i);
      if (PyBytes_Check(it)) {
        Py_INCREF(it);
      } else if (PyByteArray_Check(it)) {
        it = PyBytes_FromStringAndSize(PyByteArray_AS_STRING(it), PyByteArray_GET_SIZE(it));   // copied, like PyBackedBytes
        if (!it) { ok = false; break; }
      } else {
        PyErr_Format(PyExc_TypeError, "list element %zd: expected bytes, got %s", i, Py_TYPE(it)->tp_name);
        ok = false;
        break;
      }
      ptrs[(size_t)i] = (const uint8_t*)PyBytes_AS_STRING(it);
      lens[(size_t)i] = (uint64_t)PyBytes_GET_SIZE(it);
      done = i + 1;
    }
    if (!ok) {
      drop_range(0, done);
      std::free(chunks);
      return nullptr;
    }
    ms_extract = ms_since(t_extract);
    t_tail = std::chrono::steady_clock::now();
    const auto t_rel = std::chrono::steady_clock::now();
    Py_BEGIN_ALLOW_THREADS   // py.detach(...), src/lib.rs:82-86
    rc = rh_decode(s, ptrs, lens, (uint64_t)n, num_chunks, &opts, chunks, &out_k, want_stats ? &st : nullptr, &err);
    Py_END_ALLOW_THREADS
    ms_released = ms_since(t_rel);
    drop_range(0, n);
  }
  ms_tail = ms_since(t_tail);
  {
    DecodeProfile& pr = g_last_profile;
    pr.n = (double)n; pr.streaming = streaming ? 1 : 0; pr.alloc = ms_alloc; pr.extract = ms_extract; pr.tail = ms_tail;
    pr.total = ms_since(t_start); pr.gil_held = pr.total - ms_released;
  }
  if (pyprof)
    std::fprintf(stderr, "[pyruhvro pyprof] n=%zd streaming=%d alloc+setup=%.2f extract=%.2f engine_tail+release=%.2f total=%.2f ms\n", n,
                 (int)streaming, ms_alloc, ms_extract, ms_tail, ms_since(t_start));
  if (rc != RH_OK) {
    std::free(chunks);
    return raise_from(rc, err);
  }
  PyObject* out = PyList_New(out_k);
  for (uint32_t c = 0; c < out_k; c++) {
    ArrowArray* one = (ArrowArray*)std::malloc(sizeof(ArrowArray));
    std::memcpy(one, &chunks[c], sizeof(ArrowArray));
    PyList_SET_ITEM(out, c, PyLong_FromVoidPtr(one));
  }
  std::free(chunks);
  PyObject* stats = want_stats ? stats_dict(st) : (Py_INCREF(Py_None), Py_None);
  PyObject* ret = PyTuple_Pack(2, out, stats);
  Py_DECREF(out);
  Py_DECREF(stats);
  return ret;
}


// encode(capsule, array_addr, schema_addr, num_chunks, device=-1, stream=0, want_stats=False, kernel=0)
//   array_addr / schema_addr: ArrowArray / ArrowSchema structs the caller exported the batch's struct array into
//   (they are released here).  -> (list[int] addresses of malloc'd ArrowArray structs ("z" arrays), stats | None)
// src/lib.rs:91-106: serialize_record_batch, the GIL is released around the work like py.detach there.
PyObject* py_encode(PyObject*, PyObject* args) {
  PyObject* cap;
  unsigned long long a_addr, s_addr, num_chunks;
  int device = -1;
  unsigned long long stream = 0;
  int want_stats = 0;
  int kernel = RH_KERNEL_AUTO;
  if (!PyArg_ParseTuple(args, "OKKK|iKpi", &cap, &a_addr, &s_addr, &num_chunks, &device, &stream, &want_stats, &kernel)) return nullptr;
  ArrowArray* arr = (ArrowArray*)(uintptr_t)a_addr;
  ArrowSchema* sch = (ArrowSchema*)(uintptr_t)s_addr;
  auto drop_inputs = [&] {
    if (arr && arr->release) arr->release(arr);
    if (sch && sch->release) sch->release(sch);
  };
  rh_schema* s = get_schema(cap);
  if (!s || !arr || !sch) {
    drop_inputs();
    if (s) PyErr_SetString(PyExc_ValueError, "encode: null ArrowArray / ArrowSchema");
    return nullptr;
  }
  const uint32_t k = rh_clamp_chunks((uint64_t)arr->length, num_chunks);
  ArrowArray* chunks = (ArrowArray*)std::calloc(k, sizeof(ArrowArray));
  rh_opts opts;
  std::memset(&opts, 0, sizeof opts);
  opts.device = device;
  opts.flags = kernel;
  opts.stream = (void*)(uintptr_t)stream;
  rh_stats st;
  std::memset(&st, 0, sizeof st);
  char* err = nullptr;
  uint32_t out_k = 0;
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = rh_encode(s, arr, sch, num_chunks, &opts, chunks, &out_k, want_stats ? &st : nullptr, &err);
  Py_END_ALLOW_THREADS
  drop_inputs();
  if (rc != RH_OK) {
    std::free(chunks);
    return raise_from(rc, err);
  }
  PyObject* out = PyList_New(out_k);
  for (uint32_t c = 0; c < out_k; c++) {
    ArrowArray* one = (ArrowArray*)std::malloc(sizeof(ArrowArray));
    std::memcpy(one, &chunks[c], sizeof(ArrowArray));
    PyList_SET_ITEM(out, c, PyLong_FromVoidPtr(one));
  }
  std::free(chunks);
  PyObject* stats = want_stats ? stats_dict(st) : (Py_INCREF(Py_None), Py_None);
  PyObject* ret = PyTuple_Pack(2, out, stats);
  Py_DECREF(out);
  Py_DECREF(stats);
  return ret;
}

// release_array(addr): release (if still owned) and free an ArrowArray shell pyarrow did not consume
PyObject* py_release_array(PyObject*, PyObject* args) {
  PyObject* addr;
  if (!PyArg_ParseTuple(args, "O", &addr)) return nullptr;
  ArrowArray* a = (ArrowArray*)PyLong_AsVoidPtr(addr);
  if (!a && PyErr_Occurred()) return nullptr;
  if (a) {
    if (a->release) a->release(a);
    std::free(a);
  }
  Py_RETURN_NONE;
}

PyObject* py_device_count(PyObject*, PyObject*) { return PyLong_FromLong(rh_device_count()); }

// kernels_ready(capsule, encode=False, timeout_ms=0) -> bool: rh_schema_kernels_ready (the GIL is released while it waits)
PyObject* py_kernels_ready(PyObject*, PyObject* args) {
  PyObject* cap;
  int encode = 0;
  long timeout_ms = 0;
  if (!PyArg_ParseTuple(args, "O|pl", &cap, &encode, &timeout_ms)) return nullptr;
  rh_schema* s = get_schema(cap);
  if (!s) return nullptr;
  char* err = nullptr;
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = rh_schema_kernels_ready(s, encode, timeout_ms, &err);
  Py_END_ALLOW_THREADS
  if (rc < 0) return raise_from(RH_ERR_RUNTIME, err);
  return PyBool_FromLong(rc == 1);
}

// prebuild(capsule) -> bool (True: nothing had to be compiled): rh_schema_prebuild, GIL released
PyObject* py_prebuild(PyObject*, PyObject* args) {
  PyObject* cap;
  if (!PyArg_ParseTuple(args, "O", &cap)) return nullptr;
  rh_schema* s = get_schema(cap);
  if (!s) return nullptr;
  char* err = nullptr;
  int rc, cached = 0;
  Py_BEGIN_ALLOW_THREADS
  rc = rh_schema_prebuild(s, &cached, &err);
  Py_END_ALLOW_THREADS
  if (rc != RH_OK) return raise_from(rc, err);
  return PyBool_FromLong(cached);
}

// last_decode_profile() -> dict: phase milliseconds of this thread's most recent decode() (up to the point where the result
// list is built), incl. the time the call held the GIL
PyObject* py_last_decode_profile(PyObject*, PyObject*) {
  const DecodeProfile& pr = g_last_profile;
  return Py_BuildValue("{s:d,s:d,s:d,s:d,s:d,s:d,s:d}", "records", pr.n, "streaming", pr.streaming, "alloc_setup_ms", pr.alloc,
                       "extract_ms", pr.extract, "engine_tail_release_ms", pr.tail, "total_ms", pr.total, "gil_held_ms", pr.gil_held);
}

PyMethodDef methods[] = {
    {"compile_schema", py_compile_schema, METH_VARARGS, "compile_schema(json) -> schema capsule"},
    {"schema_ptr", py_schema_ptr, METH_VARARGS, "schema_ptr(capsule) -> int (rh_schema*)"},
    {"export_schema", py_export_schema, METH_VARARGS, "export_schema(capsule) -> address of ArrowSchema"},
    {"decode", py_decode, METH_VARARGS, "decode(capsule, list, num_chunks, device=-1, stream=0, want_stats=False, kernel=0, devices=None)"},
    {"encode", py_encode, METH_VARARGS, "encode(capsule, array_addr, schema_addr, num_chunks, device=-1, stream=0, want_stats=False, kernel=0)"},
    {"release_array", py_release_array, METH_VARARGS, "release + free an ArrowArray shell"},
    {"free_struct", py_free_struct, METH_VARARGS, "free a struct shell whose content was moved"},
    {"device_count", py_device_count, METH_NOARGS, "number of HIP devices"},
    {"kernels_ready", py_kernels_ready, METH_VARARGS, "kernels_ready(capsule, encode=False, timeout_ms=0) -> bool"},
    {"prebuild", py_prebuild, METH_VARARGS, "prebuild(capsule) -> bool"},
    {"last_decode_profile", py_last_decode_profile, METH_NOARGS, "phase milliseconds of this thread's most recent decode()"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyruhvro", "ruhvro_hip CPython boundary", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__pyruhvro(void) { return PyModule_Create(&moddef); }
