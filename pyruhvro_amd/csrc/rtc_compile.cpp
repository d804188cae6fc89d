// hiprtc front shared by libruhvro_hip.so and the compile helper rh_kcompile (see rtc_compile.h).
#include "rtc_compile.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "embedded_headers.inc"

namespace rh {

namespace {
uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
  return h;
}
}  // namespace

std::string kernel_cache_key(const std::string& source, bool encode) {
  uint64_t h = fnv1a(source);
  h = fnv1a(kHdr_program_h, h);
  h = fnv1a(kHdr_walk_h, h);
  h = fnv1a(kHdr_kernel_common_h, h);
  if (encode) {
    h = fnv1a(kHdr_encode_h, h);
    h = fnv1a(kHdr_encode_walk_h, h);
    h = fnv1a(kHdr_encode_spec_h, h);
  } else {
    h = fnv1a(kHdr_spec_body_h, h);
  }
  char buf[32];
  std::snprintf(buf, sizeof buf, "%016llx", (unsigned long long)h);
  return buf;
}

// ---------------------------------------------------------------------------
// hiprtc, loaded lazily (only a cache miss needs the compiler)
// ---------------------------------------------------------------------------
namespace {

struct Rtc {
  void* h = nullptr;
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
  std::string why;
};

Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
      r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) { r.why = std::string("hiprtc not found: ") + dlerror(); return; }
    auto sym = [&](const char* n) { return dlsym(r.h, n); };
    r.create = (decltype(r.create))sym("hiprtcCreateProgram");
    r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
    r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
    r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
    r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
    r.code = (decltype(r.code))sym("hiprtcGetCode");
    r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
    if (!r.create || !r.compile || !r.log_size || !r.log || !r.code_size || !r.code || !r.destroy) {
      r.why = "hiprtc symbols missing";
      r.h = nullptr;
    }
  });
  return r;
}

}  // namespace

std::string module_dir() {
  Dl_info info;
  if (dladdr((void*)&kernel_cache_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t s = p.rfind('/');
    if (s != std::string::npos) return p.substr(0, s);
  }
  return ".";
}

std::string kernel_cache_dir() {
  if (const char* e = std::getenv("RUHVRO_HIP_KERNEL_CACHE")) return e;
  return module_dir() + "/_kcache";
}

std::vector<char> compile_kernel(const std::string& source, std::string& log) {
  Rtc& r = rtc();
  if (!r.h) throw std::runtime_error("cannot specialise the decode kernel: " + r.why);
  const char* hdr_src[] = {kHdr_program_h, kHdr_walk_h, kHdr_kernel_common_h, kHdr_spec_body_h,
                           kHdr_encode_h, kHdr_encode_walk_h, kHdr_encode_spec_h};
  const char* hdr_name[] = {"program.h", "walk.h", "kernel_common.h", "spec_body.h", "encode.h", "encode_walk.h", "encode_spec.h"};
  void* prog = nullptr;
  if (r.create(&prog, source.c_str(), "ruhvro_spec.hip", 7, hdr_src, hdr_name) != 0)
    throw std::runtime_error("hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-w"};
  const int rc = r.compile(prog, 4, opts);
  size_t ls = 0;
  r.log_size(prog, &ls);
  if (ls > 1) {
    log.resize(ls);
    r.log(prog, &log[0]);
  }
  if (rc != 0) {
    r.destroy(&prog);
    throw std::runtime_error("hiprtc failed to compile the specialised kernel:\n" + log);
  }
  size_t sz = 0;
  r.code_size(prog, &sz);
  std::vector<char> code(sz);
  r.code(prog, code.data());
  r.destroy(&prog);
  return code;
}

}  // namespace rh
