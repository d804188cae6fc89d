// Per-device copies of a compiled schema: the schema program of the generic kernels, the specialised kernels (code objects
// from the kernel cache / the background compile jobs of kernel_jobs.h), launch helper, decode error texts.
#include "engine_internal.h"

namespace rhe {

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
const SpecKernel& spec_kernel(rh_schema* s, int device, rh::CompilePolicy policy, bool encode, bool want_fused, bool want_ranged) {
  std::map<int, std::unique_ptr<SpecKernel>>& table = encode ? s->espec : s->spec;
  SpecKernel* k = nullptr;
  {
    std::lock_guard<std::mutex> g(s->mu);
    std::unique_ptr<SpecKernel>& slot = table[device];
    if (!slot) slot.reset(new SpecKernel);
    k = slot.get();                                   // (entries are never removed while the schema lives)
    if (k->dead) return *k;
    if (k->ok && (!want_fused || k->fused_dead || k->fused_fn.load(std::memory_order_acquire)) &&
        (!want_ranged || k->ranged_dead || k->emit_r_fn.load(std::memory_order_acquire))) return *k;
  }
  const int p_size = encode ? rh::KP_ESIZE : rh::KP_SIZE, p_emit = encode ? rh::KP_EEMIT : rh::KP_EMIT;
  const unsigned parts = (1u << p_size) | (1u << p_emit) | ((want_fused && !encode) ? (1u << rh::KP_FUSED) : 0u) |
                         ((want_ranged && !encode) ? ((1u << rh::KP_SIZE_R) | (1u << rh::KP_EMIT_R)) : 0u);
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
    if (k->ok && want_ranged && !encode && !k->ranged_dead && !k->emit_r_fn.load(std::memory_order_relaxed)) {
      const rh::KernelImage& a = im[rh::KP_SIZE_R], & b = im[rh::KP_EMIT_R];
      if (a.state == rh::IMG_FAILED || a.state == rh::IMG_NONE || b.state == rh::IMG_FAILED || b.state == rh::IMG_NONE) k->ranged_dead = true;
      else if (a.state == rh::IMG_READY && b.state == rh::IMG_READY) {
        k->size_r_fn.store(load_part(*k, 3, a, rh::KP_SIZE_R), std::memory_order_release);
        k->emit_r_fn.store(load_part(*k, 4, b, rh::KP_EMIT_R), std::memory_order_release);
      }
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
                  hipEvent_t start, hipEvent_t stop) {
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


}  // namespace rhe
