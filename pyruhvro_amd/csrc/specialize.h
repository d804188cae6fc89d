// Per-schema kernel specialisation (see specialize.cpp).
#pragma once
#include <string>
#include <vector>

#include "schema.h"

namespace rh {

// Records (= threads) per workgroup of the specialised kernels: 256 unless RUHVRO_HIP_TILE says 64, 128, 512 or 1024.
int spec_tile_records();
// HIP source of the specialised k_size / k_emit pair for this schema.
std::string generate_kernel_source(const CompiledSchema& cs);
// HIP source of the specialised Arrow -> Avro pair (rh_espec_size / rh_espec_emit) for this schema.
std::string generate_encode_source(const CompiledSchema& cs);
// Content hash of (source + the device headers it includes): the on-disk cache key.
std::string kernel_cache_key(const std::string& source, bool encode = false);
// Directory of cached code objects: $RUHVRO_HIP_KERNEL_CACHE or <library dir>/_kcache.
std::string kernel_cache_dir();
// hiprtc: source -> gfx950 code object (works without a GPU).  Throws std::runtime_error.
std::vector<char> compile_kernel(const std::string& source, std::string& log);
// Cached code object, compiling (and storing) on a miss when allowed; empty if absent and !allow_compile.
// `encode` selects the Arrow -> Avro kernels instead of the decode kernels.
std::vector<char> get_kernel_image(const CompiledSchema& cs, bool allow_compile, bool* from_cache, bool encode = false);

// Host mirror of spec_body.h's spec_lds_fixed_words (LDS words in front of the window).
inline uint32_t spec_lds_fixed_words_host(int K, int nnodes, int nw, int nbm, int ndense, int nb0) {
  return (((uint32_t)(K > 0 ? K : 1) + 3) & ~3u) * (uint32_t)(nw + 1) + 128u + (uint32_t)((nnodes + 3) & ~3) + (uint32_t)((nnodes * nw + 3) & ~3) + 4 + (uint32_t)(nbm * 64) +
         (ndense > 0 ? (uint32_t)(nw * 128) : 0u) + (((uint32_t)(nb0 * nw * 2) + 3) & ~3u);
}
// domain-0 bitmap buffers of a schema (their words are collected per tile in LDS by the specialised emit kernel)
inline int dom0_bitmap_count(const CompiledSchema& cs) {
  int n = 0;
  for (const BufDesc& d : cs.bufs) n += (d.kind == BK_BITMAP && d.dom == 0) ? 1 : 0;
  return n;
}
// Lists the specialised emit kernel handles one lane per ITEM (spec_body.h dense_list): top-level arrays / maps without
// nested lists.  0 when RUHVRO_HIP_NO_DENSE=1 (A/B knob, read once: it changes the generated source and its LDS layout).
int dense_list_count(const CompiledSchema& cs);
// child-domain bitmap buffers of a schema (validity / boolean values of rows that do not line up with lanes)
inline int child_bitmap_count(const CompiledSchema& cs) {
  int n = 0;
  for (const BufDesc& d : cs.bufs) n += (d.kind == BK_BITMAP && d.dom != 0) ? 1 : 0;
  return n;
}

}  // namespace rh
