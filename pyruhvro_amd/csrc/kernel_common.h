// Workgroup-level pieces shared by the generic interpreter kernels (kernels.hip)
// and the per-schema specialised kernels (spec_body.h): chunk / workgroup
// geometry, the coalesced global -> LDS window load, lane set-up, error
// reporting and the aligned flush of staged string bytes.
#pragma once
#include "program.h"
#include "walk.h"

namespace rh {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// One workgroup = one tile = TILE (256 by default) consecutive records of ONE output chunk (chunk boundaries of
// ruhvro/src/deserialize.rs:57-68: k-1 chunks of sz rows, the last one takes the remainder).
struct Geo {
  uint32_t chunk;
  uint32_t lrow0;     // chunk-local row of lane 0 of the workgroup
  uint64_t rec0;      // global record index of lane 0
  uint32_t nrec;      // live rows in this workgroup (1..256)
};

template <int TILE = kBlock>
__device__ __forceinline__ Geo geometry(const KParams& P, uint32_t b) {
  Geo g;
  uint32_t chunk = b / P.bpc;
  if (chunk > P.k - 1) chunk = P.k - 1;
  const uint32_t lb = b - chunk * P.bpc;
  const uint64_t rows_c = chunk == P.k - 1 ? P.rows_last : P.sz;
  g.chunk = chunk;
  g.lrow0 = lb * TILE;
  g.rec0 = (uint64_t)chunk * P.sz + g.lrow0;
  const uint64_t left = rows_c - g.lrow0;
  g.nrec = left < (uint64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
  return g;
}

// XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md), and
// each XCD has its own L2.  Neighbouring tiles write neighbouring bytes of every output buffer (a validity
// bitmap advances 32 bytes per tile), so giving each XCD one CONTIGUOUS run of tiles lets its L2 merge those
// partial lines instead of eight L2s each owning a sliver.  Pure performance mapping: any placement is correct.
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t nblocks) {
  const uint32_t per = nblocks >> 3;
  return b < (per << 3) ? (b & 7u) * per + (b >> 3) : b;
}

// One straight-line round of the window load: up to KB rows of TILE 16-byte vectors.  Every load is issued
// before the first LDS store.  Whole rows carry no per-lane predicate (the row count is wave-uniform); only the
// last, partial row is predicated.  Deliberately NOT a loop: with a loop the register array is loop-carried and
// the compiler protects each element with an s_waitcnt vmcnt(0) BEFORE re-loading it, which serialises every
// load behind the previous one's HBM round trip (seen in the ISA; it made this phase 13.5k cycles per wave).
// Exactly ROWS unpredicated rows + the partial row, selected by a scalar switch on the row count: the executed path
// is ~2 instructions per row instead of the ~24 of a per-row scalar row test + last-row lane test
// (k_emit 0.950 -> 0.940 ms, profiles/r02a_variants_ab.txt).
template <int TILE, int ROWS>
__device__ __forceinline__ void stage_exact(const RH_GLOBAL v4u* gp, v4u* lp, uint32_t rem, uint32_t tid) {
#ifndef RH_V_NODMA
  // The rows are moved by LDS-DMA -- global_load_lds_dwordx4: 1 KiB per wave instruction, destination = a wave-uniform
  // LDS base (M0) + lane x 16 -- with no VGPR round trip and no ds_write_b128 (13 LDS cycles each, eight per thread: 16 %
  // of the size kernel's LDS instructions): k_size -3.4 %, k_emit -1 % (profiles/r04g_lds_dma_ab.txt; RH_V_NODMA = the
  // register path below, kept for the A/B).  stage_window waits vmcnt(0) for them in front of the workgroup barrier.
  {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    RH_LDS uint8_t* const lw = (RH_LDS uint8_t*)(lp - tid) + wave * 1024u;                 // this wavefront's 1 KiB of row 0
    const RH_GLOBAL uint8_t* const g0 = reinterpret_cast<const RH_GLOBAL uint8_t*>(gp);    // this lane's 16 bytes of row 0
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      __builtin_amdgcn_global_load_lds(g0 + (size_t)j * (TILE * 16u), lw + j * (TILE * 16), 16, 0, 0);
    if (tid < rem) __builtin_amdgcn_global_load_lds(g0 + (size_t)ROWS * (TILE * 16u), lw + ROWS * (TILE * 16), 16, 0, 0);
    return;
  }
#endif
  // uniform base + 32-bit lane offset: the loads take the scalar-base addressing form, one v_add per row
  const uintptr_t base = reinterpret_cast<uintptr_t>(gp - tid);
  const uint32_t voff = tid * 16u;
  v4u r[ROWS + 1];
#pragma unroll
  for (int j = 0; j < ROWS; j++) r[j] = *reinterpret_cast<const RH_GLOBAL v4u*>(base + (voff + (uint32_t)j * (TILE * 16u)));
  const bool part = tid < rem;
  if (part) r[ROWS] = *reinterpret_cast<const RH_GLOBAL v4u*>(base + (voff + (uint32_t)ROWS * (TILE * 16u)));
#pragma unroll
  for (int j = 0; j < ROWS; j++) lp[j * TILE] = r[j];
  if (part) lp[ROWS * TILE] = r[ROWS];
}

template <int TILE, int KB>
__device__ __forceinline__ void stage_rows(const RH_GLOBAL v4u* gp, v4u* lp, uint32_t left, uint32_t tid) {
  const uint32_t rows = left / TILE;            // wave-uniform
  const uint32_t rem = left - rows * TILE;
  static_assert(KB == 12, "the switch below lists 0..12 rows");
  switch (__builtin_amdgcn_readfirstlane(rows)) {
    case 0: stage_exact<TILE, 0>(gp, lp, rem, tid); break;
    case 1: stage_exact<TILE, 1>(gp, lp, rem, tid); break;
    case 2: stage_exact<TILE, 2>(gp, lp, rem, tid); break;
    case 3: stage_exact<TILE, 3>(gp, lp, rem, tid); break;
    case 4: stage_exact<TILE, 4>(gp, lp, rem, tid); break;
    case 5: stage_exact<TILE, 5>(gp, lp, rem, tid); break;
    case 6: stage_exact<TILE, 6>(gp, lp, rem, tid); break;
    case 7: stage_exact<TILE, 7>(gp, lp, rem, tid); break;
    case 8: stage_exact<TILE, 8>(gp, lp, rem, tid); break;
    case 9: stage_exact<TILE, 9>(gp, lp, rem, tid); break;
    case 10: stage_exact<TILE, 10>(gp, lp, rem, tid); break;
    case 11: stage_exact<TILE, 11>(gp, lp, rem, tid); break;
    default: stage_exact<TILE, 12>(gp, lp, 0u, tid); break;     // left <= KB * TILE: 12 full rows, no partial one
  }
}

// Stage [wb16, we) of the payload into LDS with 16-byte loads (1 KiB per wave instruction): one HBM round trip
// for windows up to 12 rows (48 KiB at 256 threads), two for the largest windows the host ever configures (96 KiB).
template <int TILE = kBlock>
__device__ __forceinline__ void stage_window(const KParams& P, uint8_t* win, uint64_t wb16, uint64_t we, uint32_t tid) {
  const uint32_t nvec = (uint32_t)((we - wb16 + 15) >> 4);
  // vectors that lie completely inside the payload (only the very last tile of a call can have a ragged one)
  const uint32_t nfull = (uint32_t)(wb16 + ((uint64_t)nvec << 4) <= P.data_len ? nvec : (P.data_len - wb16) >> 4);
  constexpr int KB = 12;
  const RH_GLOBAL v4u* gp = reinterpret_cast<const RH_GLOBAL v4u*>(reinterpret_cast<uintptr_t>(P.data + wb16)) + tid;
  v4u* lp = reinterpret_cast<v4u*>(win) + tid;
  const uint32_t first = nfull < (uint32_t)(KB * TILE) ? nfull : (uint32_t)(KB * TILE);
  stage_rows<TILE, KB>(gp, lp, first, tid);
  if (nfull > (uint32_t)(KB * TILE)) {
    uint32_t done = KB * TILE;
    while (done < nfull) {                        // windows beyond 48 KiB (at 256 threads): further rounds
      const uint32_t left = nfull - done < (uint32_t)(KB * TILE) ? nfull - done : (uint32_t)(KB * TILE);
      stage_rows<TILE, KB>(gp + done, lp + done, left, tid);
      done += left;
    }
  }
#ifndef RH_V_NODMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA rows have landed before this wave reaches the barrier
#endif
  if (nfull < nvec && tid < 16) {   // ragged tail of the payload: bytes, zero-filled past the end
    const uint64_t pos = wb16 + ((uint64_t)nfull << 4);
    win[(nfull << 4) + tid] = pos + tid < P.data_len ? P.data[pos + tid] : 0;
  }
}

// lane set-up from record offsets the caller already holds in registers (loaded once, together with
// the window bounds, so the workgroup pays a single dependent round trip before the window load)
__device__ __forceinline__ void lane_init_from(Lane& L, const Geo& g, uint64_t o0, uint64_t o1, uint64_t wb16, uint32_t tid) {
  L.live = tid < g.nrec;
  L.pres = L.live;
  L.err = 0;
  L.edetail = 0;
  L.redo = false;
  L.pstk = 0; L.lstk = 0; L.sstk = 0; L.la = 0;
  L.cur = L.live ? (uint32_t)(o0 - wb16) : 0;
  L.end = L.live ? (uint32_t)(o1 - wb16) : 0;
}

__device__ __forceinline__ void lane_init(Lane& L, const KParams& P, const Geo& g, uint64_t wb16, uint32_t tid) {
  L.live = tid < g.nrec;
  L.pres = L.live;
  L.err = 0;
  L.edetail = 0;
  L.redo = false;
  L.pstk = 0; L.lstk = 0; L.sstk = 0; L.la = 0;
  L.cur = 0; L.end = 0;
  if (L.live) {
    const uint64_t o0 = P.offsets[g.rec0 + tid], o1 = P.offsets[g.rec0 + tid + 1];
    L.cur = (uint32_t)(o0 - wb16);
    L.end = (uint32_t)(o1 - wb16);
  }
}

// Lowest erroring lane of the workgroup reports (code, detail); lowest record index wins globally
// (== the in-order join of deserialize.rs:115-119 + first `?` in fast_decode.rs:827).
// Contains a workgroup barrier.
__device__ __forceinline__ void report_errors(const KParams& P, uint32_t* misc, const Lane& L, const Geo& g, uint32_t tid, uint32_t tile) {
  if (L.err) atomicMin(&misc[0], tid);
  __syncthreads();
  if (misc[0] == tid) {
    ErrInfo ei; ei.code = L.err; ei.pad = 0; ei.detail = L.edetail;
    P.errinfo[tile] = ei;
    atomicMax(P.first_bad, ~(unsigned long long)(g.rec0 + tid));
  }
}

}  // namespace rh
