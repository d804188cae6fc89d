// Per-schema kernel specialisation (see specialize.cpp).
#pragma once
#include <string>
#include <vector>

#include "schema.h"

namespace rh {

// Records (= threads) per workgroup of the specialised kernels: 256 unless RUHVRO_HIP_TILE says 64, 128, 512 or 1024.
int spec_tile_records();
// ... of THIS schema: one wavefront (program.h kWideTile) when it is wide, else the above
int spec_tile_records(const CompiledSchema& cs);
// The specialised kernels of a schema.  Each is its own hiprtc program (its own code object in the kernel cache), so that a
// cache miss compiles them side by side (kernel_jobs.cpp) and the opt-in single-pass kernel is only built when asked for.
// (KP_SIZE_R / KP_EMIT_R, round 6: the RANGED pair -- the same passes for the tiles that do not fit the LDS window, spec_body.h
//  ranged_tile; compiled when a schema first meets such tiles, launched behind the pair above from then on)
enum KernelPart { KP_SIZE = 0, KP_EMIT = 1, KP_FUSED = 2, KP_ESIZE = 3, KP_EEMIT = 4, KP_SIZE_R = 5, KP_EMIT_R = 6, KP_COUNT = 7 };
constexpr unsigned kDecodeParts = 7u | 96u, kEncodeParts = 24u;
inline bool kernel_part_is_encode(int part) { return part == KP_ESIZE || part == KP_EEMIT; }
inline const char* kernel_part_entry(int part) {
  static const char* const names[KP_COUNT] = {"rh_spec_size", "rh_spec_emit", "rh_spec_fused", "rh_espec_size", "rh_espec_emit", "rh_spec_size_r", "rh_spec_emit_r"};
  return names[part];
}
// HIP source of the specialised decode kernels (rh_spec_size / rh_spec_emit / rh_spec_fused) of this schema; `parts` = the
// kernel entries to emit (bit KernelPart).  The default is the whole set: what rh_schema_kernel_source shows and what the
// measurement stamps hash (rh_schema_kernel_key).
std::string generate_kernel_source(const CompiledSchema& cs, unsigned parts = kDecodeParts);
// HIP source of the specialised Arrow -> Avro pair (rh_espec_size / rh_espec_emit) for this schema.
std::string generate_encode_source(const CompiledSchema& cs, unsigned parts = kEncodeParts);
// Source of ONE part, or "" when the schema has no such kernel (rh_spec_fused needs K <= 64; the encode pair a schema
// rh_encode takes).
std::string generate_part_source(const CompiledSchema& cs, int part);

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
