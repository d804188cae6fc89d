// Code objects of a schema's specialised kernels: the on-disk cache and the compile jobs behind it.
//
// The reference pays a JSON parse for a schema it has not seen (src/lib.rs:39-54, deserialize.rs:18-20); a hiprtc
// compile of the specialised kernels is tens of seconds.  So a cache miss never blocks a decode call: the call is served
// by the generic interpreter kernels (kernels.hip, same handlers, same buffers) while every missing kernel is compiled
// in the background -- each kernel its own hiprtc program, each program in its own `rh_kcompile` helper process
// (kcompile_main.cpp: real parallelism, nothing of LLVM in the caller's address space, killed at exit), or on a thread
// of this process when the helper is not installed.  The next call after the jobs finish loads the code objects.
// No lock of the schema is held while anything compiles.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "specialize.h"

namespace rh {

enum CompilePolicy {
  CP_CACHED_ONLY = 0,   // disk cache or nothing (small calls: not worth a compile)
  CP_BACKGROUND = 1,    // a miss starts the compile jobs and returns IMG_COMPILING
  CP_BLOCKING = 2       // a miss starts the jobs and waits for them (RH_KERNEL_SPECIALIZED, rh_schema_prebuild)
};
enum ImageState { IMG_UNKNOWN = -1, IMG_READY = 0, IMG_NOT_CACHED = 1, IMG_COMPILING = 2, IMG_FAILED = 3, IMG_NONE = 4 /* the schema has no such kernel */ };

struct KernelImage {
  ImageState state = IMG_UNKNOWN;
  std::shared_ptr<const std::vector<char>> code;
  std::string why;            // IMG_FAILED: the compiler's log / the reason
  bool from_cache = false;    // read from the disk cache (not compiled by this process)
  double seconds = 0;         // wall time of the compile job (0 for a cache hit)
};

class KernelImages;            // per schema, device independent; outlives the schema while a job runs
std::shared_ptr<KernelImages> new_kernel_images();

// The images of `parts` (bit KernelPart) in out[part]; missing ones are looked up on disk and, as the policy allows,
// compiled -- all of them started before any is waited for.  Returns the number of compile jobs this call started.
unsigned kernel_images(const std::shared_ptr<KernelImages>& im, const CompiledSchema& cs, unsigned parts, CompilePolicy policy,
                       KernelImage out[KP_COUNT]);
// Wait (at most timeout_ms; < 0 = no limit) until none of `parts` is IMG_COMPILING.  1 = all ready (or IMG_NONE), 0 = some
// still compiling / not cached / never asked for, -1 = one failed (*why).
int kernel_images_wait(const std::shared_ptr<KernelImages>& im, unsigned parts, long timeout_ms, std::string* why);

}  // namespace rh
