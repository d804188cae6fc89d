// hiprtc front: generated HIP source -> gfx950 code object, and the content-hash key of the kernel cache.
// Shared by the library (a synchronous compile, rh_schema_prebuild) and by the out-of-process compile helper
// `rh_kcompile` (kcompile_main.cpp) that the background compile jobs of kernel_jobs.cpp run.
#pragma once
#include <string>
#include <vector>

namespace rh {

// Content hash of (source + the device headers it includes): the on-disk cache key.
std::string kernel_cache_key(const std::string& source, bool encode = false);
// Directory of cached code objects: $RUHVRO_HIP_KERNEL_CACHE or <library dir>/_kcache.
std::string kernel_cache_dir();
// Directory this code was loaded from (the library's, or the helper executable's).
std::string module_dir();
// hiprtc: source -> gfx950 code object (works without a GPU).  Throws std::runtime_error.
std::vector<char> compile_kernel(const std::string& source, std::string& log);

}  // namespace rh
