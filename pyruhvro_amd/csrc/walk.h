// Device-side record walker of the Avro -> Arrow direct-decode path (gfx950).
//
// One lane = one record (the per-record unit of the reference's hot loop,
// ruhvro/src/fast_decode.rs:825-828).  Control flow is WAVE-UNIFORM: all 64
// lanes execute the same field handler, and the reference's data-dependent
// branches become per-lane predicates
//   live -- the lane owns a row in the current row domain,
//   pres -- the row is decoded from bytes (FieldDecoder::decode, 421-499) or
//           null-filled (FieldDecoder::append_null, 503-534).
// The handlers below (h_fixed, h_string, h_rec_*, h_union_*, h_list_*) are the
// single statement of those semantics.  They are used twice:
//   * by the generic interpreter in kernels.hip (Op fetched at run time),
//   * by per-schema specialised kernels (specialize.cpp emits a straight-line
//     call sequence with constexpr Ops, so everything below constant-folds).
// Each handler is a template on the walk mode (EMIT = false: size pass,
// counters only; EMIT = true: materialise Arrow buffers), the byte source
// (LDS window or global memory) and the context type that says where the
// per-lane counters live (LDS for the interpreter, registers when specialised).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "program.h"

namespace rh {

typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;
typedef uint32_t v4w __attribute__((ext_vector_type(4)));
typedef v4w __attribute__((aligned(1))) v4wu;

// --------------------------------------------------------------------------
// wave primitives (wave = 64 lanes)
// --------------------------------------------------------------------------
// Inclusive wave scan in the VALU with DPP (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast
// 15 and 31 across rows): six v_add with DPP modifiers, no LDS-crossbar (ds_bpermute) round trips.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t /*lane*/) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return v;
}

// wave total, returned wave-uniform (SGPR)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}

// --------------------------------------------------------------------------
// byte sources.  Positions are relative to the workgroup's window base.
//
// Measured on MI355X (tools/ldslat.hip): a DS read that is not naturally aligned (ds_read_b64 off an
// 8-byte boundary, ds_read_b32 off a 4-byte boundary) is serviced one lane per cycle -- ~64 LDS cycles
// per wave instruction, serialised across every wave of the CU (128 cycles/step for one wave, >1000 with
// 16 waves) -- while aligned reads pipeline at ~64 cycles latency regardless of load.  So the LDS window
// is only ever read with ALIGNED dword reads and the bytes are funnelled into place with v_alignbyte.
// Global memory has no such cliff (tools/gmemalign.hip), so GlobalSrc reads unaligned directly.
// --------------------------------------------------------------------------
struct LdsSrc {
  static constexpr bool kSlide = false;
  const uint8_t* w;   // LDS window, 16-byte aligned; >= 12 readable bytes past the last record
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const { return w[p]; }
  // 8 bytes at any byte position: three aligned dwords (ds_read2_b32 + ds_read_b32), two v_alignbyte
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
    const uint32_t sh = p & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
  }
  // 12 bytes at any byte position (a branch byte + a 10-byte varint): four aligned dwords (two ds_read2_b32), three v_alignbyte
  __device__ __forceinline__ void ld12(uint32_t p, uint64_t& lo, uint32_t& hi) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3];
    const uint32_t sh = p & 3u;
    lo = ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d3, d2, sh);
  }
  // 16 bytes at any byte position: five aligned dwords, four v_alignbyte
  __device__ __forceinline__ v4w ld16(uint32_t p) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], d4 = a[4];
    const uint32_t sh = p & 3u;
    v4w r;
    r.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
    r.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
    r.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
    return r;
  }
  // 4 bytes at any byte position from ONE ds_read2_b32 and one v_alignbyte
  __device__ __forceinline__ uint32_t ld4(uint32_t p) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    return __builtin_amdgcn_alignbyte(a[1], a[0], p & 3u);
  }
  // 16 bytes at a 16-byte aligned position: one ds_read_b128
  static constexpr bool kAligned16 = true;
  __device__ __forceinline__ v4w ld16a(uint32_t p) const { return *reinterpret_cast<const v4w*>(w + p); }
  // >= 5 valid bytes (a 1-byte union branch + a varint of <= 4 bytes) from ONE ds_read2_b32
  __device__ __forceinline__ uint64_t ld5(uint32_t p) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    const uint32_t d0 = a[0], d1 = a[1];
    const uint32_t sh = p & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(0u, d1, sh);
    return ((uint64_t)hi << 32) | lo;
  }
  // copy8_pieces: the aligned dword that holds byte p, and the two aligned dwords behind byte p + j (j a multiple of 4)
  __device__ __forceinline__ uint32_t ld4a(uint32_t p) const { return *reinterpret_cast<const uint32_t*>(w + (p & ~3u)); }
  __device__ __forceinline__ void next2(uint32_t p, int j, uint32_t& d1, uint32_t& d2) const {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(w + (p & ~3u));
    d1 = a[j / 4 + 1]; d2 = a[j / 4 + 2];
  }
};

// The same window, addressed ABSOLUTELY: positions are LDS byte addresses (the window's own LDS address is added to
// the cursors once, when a lane starts).  The address of dynamic LDS is only known to the backend after instruction
// selection, so `window + position` costs every read two VALU adds (`+ 0` for the symbol, `+ the window's offset`, too
// large for the offset field of ds_read2_b32) that an absolute position does not need: ~8 % of the size walk's VALU.
#ifndef RH_LDS
#define RH_LDS __attribute__((address_space(3)))
#endif
#define RH_GLOBAL __attribute__((address_space(1)))
struct LdsAbsSrc {
  static constexpr bool kSlide = false;
  static __device__ __forceinline__ const RH_LDS uint32_t* dw(uint32_t p) { return reinterpret_cast<const RH_LDS uint32_t*>((uintptr_t)(p & ~3u)); }
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const { return *reinterpret_cast<const RH_LDS uint8_t*>((uintptr_t)p); }
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const {
    const RH_LDS uint32_t* a = dw(p);
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
    const uint32_t sh = p & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
  }
  __device__ __forceinline__ void ld12(uint32_t p, uint64_t& lo, uint32_t& hi) const {
    const RH_LDS uint32_t* a = dw(p);
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3];
    const uint32_t sh = p & 3u;
    lo = ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d3, d2, sh);
  }
  __device__ __forceinline__ v4w ld16(uint32_t p) const {
    const RH_LDS uint32_t* a = dw(p);
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], d4 = a[4];
    const uint32_t sh = p & 3u;
    v4w r;
    r.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
    r.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
    r.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
    return r;
  }
  __device__ __forceinline__ uint32_t ld4(uint32_t p) const {
    const RH_LDS uint32_t* a = dw(p);
    return __builtin_amdgcn_alignbyte(a[1], a[0], p & 3u);
  }
  __device__ __forceinline__ uint64_t ld5(uint32_t p) const {
    const RH_LDS uint32_t* a = dw(p);
    const uint32_t d0 = a[0], d1 = a[1];
    const uint32_t sh = p & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(0u, d1, sh);
    return ((uint64_t)hi << 32) | lo;
  }
  __device__ __forceinline__ uint32_t ld4a(uint32_t p) const { return dw(p)[0]; }
  __device__ __forceinline__ void next2(uint32_t p, int j, uint32_t& d1, uint32_t& d2) const {
    const RH_LDS uint32_t* a = dw(p);
    d1 = a[j / 4 + 1]; d2 = a[j / 4 + 2];      // (ds_read2_b32 with immediate offsets off one address register)
  }
  static constexpr bool kAligned16 = true;
  __device__ __forceinline__ v4w ld16a(uint32_t p) const { return *reinterpret_cast<const RH_LDS v4w*>((uintptr_t)p); }      // one ds_read_b128
};

struct GlobalSrc {
  static constexpr bool kSlide = false;
  const uint8_t* g;   // payload + window base
  uint64_t lim;       // readable bytes from g
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const { return (uint64_t)p < lim ? g[p] : 0u; }
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const {
    if ((uint64_t)p + 8 <= lim) return *reinterpret_cast<const u64u*>(g + p);
    uint64_t x = 0;
    for (uint32_t j = 0; j < 8 && (uint64_t)p + j < lim; j++) x |= (uint64_t)g[p + j] << (8 * j);
    return x;
  }
  __device__ __forceinline__ uint64_t ld5(uint32_t p) const { return ld8(p); }
  __device__ __forceinline__ uint32_t ld4(uint32_t p) const { return (uint32_t)ld8(p); }
  __device__ __forceinline__ void ld12(uint32_t p, uint64_t& lo, uint32_t& hi) const {
    if ((uint64_t)p + 12 <= lim) { lo = *reinterpret_cast<const u64u*>(g + p); hi = *reinterpret_cast<const u32u*>(g + p + 8); return; }
    lo = ld8(p);
    hi = 0;
    for (uint32_t j = 8; j < 12 && (uint64_t)p + j < lim; j++) hi |= (uint32_t)g[p + j] << (8 * (j - 8));
  }
  // (copy8_pieces over global memory: unaligned reads are fine here, so the "aligned dwords" are taken as if p & 3 were 0
  //  -- the pieces' funnel shift is by p & 3, which callers of a GlobalSrc get right by reading at p - (p & 3))
  __device__ __forceinline__ uint32_t ld4a(uint32_t p) const { return (uint32_t)ld8(p & ~3u); }
  __device__ __forceinline__ void next2(uint32_t p, int j, uint32_t& d1, uint32_t& d2) const {
    const uint64_t x = ld8((p & ~3u) + (uint32_t)j + 4u);
    d1 = (uint32_t)x; d2 = (uint32_t)(x >> 32);
  }
  __device__ __forceinline__ v4w ld16(uint32_t p) const {
    const uint64_t lo = ld8(p), hi = ld8(p + 8);
    v4w r;
    r.x = (uint32_t)lo; r.y = (uint32_t)(lo >> 32); r.z = (uint32_t)hi; r.w = (uint32_t)(hi >> 32);
    return r;
  }
  static constexpr bool kAligned16 = false;     // (global memory has no alignment cliff: ld16 is the read)
  __device__ __forceinline__ v4w ld16a(uint32_t p) const { return ld16(p); }
};

// One WAVEFRONT stages `nbytes` (a multiple of 16) from global `g` (16-byte aligned) to LDS address `wa` by LDS-DMA: 1 KiB per
// instruction (global_load_lds_dwordx4: M0 = a wave-uniform LDS base, lane x 16 behind it), all rows in flight together, then
// vmcnt(0).  No workgroup barrier: for a window only this wavefront reads (SlideSrc::refill).  `gvalid` = bytes readable at g:
// vectors that would cross it are zero-filled / copied by bytes.
__device__ __forceinline__ void stage_wave(const uint8_t* g, uint64_t gvalid, uint32_t wa, uint32_t nbytes, uint32_t lane) {
  const uint32_t nfull = (uint64_t)nbytes <= gvalid ? nbytes : (uint32_t)(gvalid & ~15ull);      // bytes in whole vectors inside the payload
  const RH_GLOBAL uint8_t* const g0 = reinterpret_cast<const RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(g)) + lane * 16u;
  for (uint32_t off = 0; off < nfull; off += 1024u) {
    RH_LDS uint8_t* const lw = reinterpret_cast<RH_LDS uint8_t*>((uintptr_t)(wa + off));
    if (off + lane * 16u < nfull) __builtin_amdgcn_global_load_lds(g0 + off, lw, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (nfull < nbytes) {                      // the ragged end of the payload: bytes, zero behind them
    for (uint32_t i = nfull + lane; i < nbytes; i += 64u)
      *reinterpret_cast<RH_LDS uint8_t*>((uintptr_t)(wa + i)) = (uint64_t)i < gvalid ? g[i] : (uint8_t)0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The same by WHATEVER lanes of the wavefront are active.  SlideSrc::refill runs at the head of every list iteration of a
// record larger than the window, and the compiler is free to execute such a loop with only the lanes that still have items
// enabled (observed: EXEC = the one live lane inside the block loops of an array of arrays, profiles/r06_o_*): the LDS-DMA form
// above then stages one 16-byte piece per KiB and leaves the rest of the window stale.  All lanes active: the DMA form; else
// the active lanes deal the pieces among themselves (through registers: an LDS-DMA lands at M0 + lane id x 16).
__device__ __forceinline__ void stage_wave_any(const uint8_t* g, uint64_t gvalid, uint32_t wa, uint32_t nbytes, uint32_t lane) {
  const uint64_t ex = __ballot(true);
  if (ex == ~0ull) { stage_wave(g, gvalid, wa, nbytes, lane); return; }
  const uint32_t cnt = (uint32_t)__popcll(ex);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u));
  const uint32_t nfull = (uint64_t)nbytes <= gvalid ? nbytes : (uint32_t)(gvalid & ~15ull);
  for (uint32_t o = rank * 16u; o < nfull; o += cnt * 16u) {
    const v4w x = *reinterpret_cast<const v4w*>(g + o);
    *reinterpret_cast<RH_LDS v4w*>((uintptr_t)(wa + o)) = x;
  }
  for (uint32_t i = nfull + rank; i < nbytes; i += cnt)
    *reinterpret_cast<RH_LDS uint8_t*>((uintptr_t)(wa + i)) = (uint64_t)i < gvalid ? g[i] : (uint8_t)0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// n bytes (a multiple of 16, per lane) from global memory to this lane's slice of the LDS window (SlideSrc lane windows).  A real
// function: it is reached from every top-up hook of a wide schema's walk; its arguments travel in registers.
__device__ __attribute__((noinline)) void stage_lane_copy(const RH_GLOBAL uint8_t* gs, uint32_t lds, uint32_t n) {
  for (uint32_t o = 0; o < n; o += 64u) {      // four vectors in flight per round
    v4w x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (o + 16u * k < n) x[k] = *reinterpret_cast<const RH_GLOBAL v4w*>(gs + o + 16u * k);
#pragma unroll
    for (int k = 0; k < 4; k++) if (o + 16u * k < n) *reinterpret_cast<RH_LDS v4w*>((uintptr_t)(lds + o + 16u * k)) = x[k];
  }
}

// Round 6: the source of every tile that does not fit the LDS window in one piece.  Positions are LDS byte addresses like
// LdsAbsSrc's (window byte 0 = `wa`), but only the first `wlen` bytes behind `wa` are staged: a read that is not completely
// inside them is served from global memory at the same offset from `g` (per lane; positions in front of the window -- a lane
// that stayed behind a refill -- included).  Such tiles are processed in RANGES of consecutive records (spec_body.h
// ranged_tile): a range of records that fits is staged whole (no read ever leaves the window); a single record LARGER than
// the window (`sliding`) starts with its first bytes staged and, at every list iteration, moves the window up to its cursor
// (refill) -- the lone lane that owns such a record reads LDS at ~64 cycles per dependent head instead of HBM at ~1500, which
// was the whole cost of a giant record (profiles/r05t_*: ~1400 cycles per item).
struct SlideSrc {
  static constexpr bool kSlide = true;
  static constexpr bool kMoves = true;      // (its window can move: refill)
  uint32_t wa;            // LDS address of window byte 0 (wave-uniform, 16-byte aligned)
  mutable uint32_t wlen;  // staged bytes behind wa (wave-uniform; zero-filled past the end of the payload)
  uint32_t wcap;          // bytes the window can hold (a multiple of 16)
  mutable const uint8_t* g;   // global address of window byte 0 (16-byte aligned)
  mutable uint64_t glim;  // readable bytes behind g
  bool sliding;           // a single record larger than the window: refill() moves the window
#ifdef RH_WIDE_SCHEMA
  // LANE WINDOWS (wide schemas' direct tiles, spec_body.h ranged_tile): every lane walks its own record of a kilobyte and more,
  // too large for a range to hold many of them.  Each lane stages the next `lwin` bytes of ITS record into a slice of the window
  // of its own (lw; slices 16 bytes apart in their bank phase) and tops it up at the hooks the generator places between the
  // columns (h_topup) -- a record's ~400 dependent global reads become ~10 refills of 16 independent loads.  Positions stay
  // the tile-wide coordinates of the direct walk (wa + payload offset - rb16: dense_list's table works across lanes), so a
  // position maps to LDS through the lane's own (p0, lw), and whatever is not in the slice is read from global memory as before.
  bool lanes = false;
  uint32_t lwin = 0;          // bytes of a lane's slice
  uint32_t lw = 0;            // LDS address of this lane's slice
  mutable uint32_t p0 = 0;    // position of the slice's first byte
  __device__ __forceinline__ uint32_t base() const { return lanes ? p0 : wa; }
  __device__ __forceinline__ uint32_t lds(uint32_t p) const { return lanes ? p - p0 + lw : p; }
#else
  static constexpr bool lanes = false;
  __device__ __forceinline__ void stage_lane() const {}
  __device__ __forceinline__ uint32_t base() const { return wa; }
  __device__ __forceinline__ uint32_t lds(uint32_t p) const { return p; }
#endif
  __device__ __forceinline__ bool in(uint32_t p, uint32_t need) const { return p - base() + need <= wlen; }      // (unsigned: false in front of the window)
  // Move the window up to the cursor of the one live lane (`owner`) once it has used half of it (called at every list
  // iteration / dense round).  Returns the distance every position has to be rebased by (0: not moved); positions noted before
  // the call (dense_list's table) must have been used up.  g / glim / wlen are per-lane copies of wave-level state, and a lane
  // that was masked off during an earlier call (see stage_wave_any) kept an older base: the OWNER's copy is the authoritative
  // one -- it takes part in every call, since a call without it finds no live lane -- and every call starts from it.
  __device__ __forceinline__ void sync(int owner) const {
    const uint64_t ga = (uint64_t)reinterpret_cast<uintptr_t>(g);
    const uint64_t go = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ga >> 32), owner) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ga, owner);
    g = reinterpret_cast<const uint8_t*>((uintptr_t)go);
    glim = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(glim >> 32), owner) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)glim, owner);
    wlen = (uint32_t)__builtin_amdgcn_readlane((int)wlen, owner);
  }
  __device__ __forceinline__ uint32_t advance_to(uint32_t cur, uint32_t lane, int owner) const {
    if (!sliding) return 0;
    const uint32_t off = cur - wa;
    if ((int32_t)off < (int32_t)(wcap / 2)) return 0;
    sync(owner);
    const uint32_t delta = off & ~15u;
    g += delta;
    glim -= delta;
    const uint64_t left = glim + 15ull;
    wlen = left < (uint64_t)wcap ? (uint32_t)(left & ~15ull) : wcap;
    stage_wave_any(g, glim, wa, wlen, lane);
    return delta;
  }
  template <class LaneT>
  __device__ __forceinline__ void refill(LaneT& L, uint32_t lane) const {
#ifdef RH_WIDE_SCHEMA
    if (lanes) {      // every live lane that has moved on: its slice from its cursor on -- when ANY lane runs low (the lanes of a wave stay in step)
      const uint32_t off = L.cur - p0;
      const bool low = L.live && (int32_t)(wlen - off) < (int32_t)kLaneLow;
      if (!__any(low)) return;
      if (L.live && off >= 16u) {
        p0 += off & ~15u;
        stage_lane();
      }
      return;
    }
#endif
    if (!sliding) return;
    const uint64_t lv = __ballot(L.live);
    if (lv == 0 || (lv & (lv - 1)) != 0) return;                      // (one record per sliding range: one live lane)
    const int owner = (int)__builtin_ctzll(lv);
    sync(owner);                                                      // (lanes that sat out earlier calls catch up: dense_list's item lanes, copy_bytes_coop)
    const uint32_t delta = advance_to((uint32_t)__builtin_amdgcn_readlane((int)L.cur, owner), lane, owner);
    L.cur -= delta;
    L.end -= delta;
  }
#ifdef RH_WIDE_SCHEMA
  static constexpr uint32_t kLaneLow = 64;      // refill when a lane has fewer staged bytes than this ahead of its cursor
  // this lane's slice <- the lwin bytes of the payload from position p0 (whole 16-byte vectors inside the payload; what is left
  // of a ragged end is read from global memory)
  __device__ __forceinline__ void stage_lane() const {
    const uint64_t goff = (uint64_t)(p0 - wa);
    const uint64_t left = glim > goff ? glim - goff : 0ull;
    const uint32_t n = left >= (uint64_t)lwin ? lwin : (uint32_t)(left & ~15ull);
    stage_lane_copy(reinterpret_cast<const RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(g)) + goff, lw, n);
    wlen = n;
  }
#endif
  __device__ __forceinline__ GlobalSrc far(uint32_t p, uint32_t& q) const {
    // the same position as an offset from a global base that is valid for this lane: g moved forward by refills, the lane may not have
    const int64_t off = (int64_t)(int32_t)(p - wa);
    q = 0;
    // (a lane of the fast walk that lost its record may hold a cursor far behind the payload: nothing is readable there)
    return GlobalSrc{g + off, off < (int64_t)glim ? (uint64_t)((int64_t)glim - off) : 0ull};
  }
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const {
    if (in(p, 1)) return LdsAbsSrc().ld1(lds(p));
    uint32_t q; const GlobalSrc f = far(p, q); return f.ld1(q);
  }
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const {
    if (in(p, 12)) return LdsAbsSrc().ld8(lds(p));
    uint32_t q; const GlobalSrc f = far(p, q); return f.ld8(q);
  }
  __device__ __forceinline__ uint64_t ld5(uint32_t p) const { return ld8(p); }
  __device__ __forceinline__ uint32_t ld4(uint32_t p) const {
    if (in(p, 8)) return LdsAbsSrc().ld4(lds(p));
    uint32_t q; const GlobalSrc f = far(p, q); return f.ld4(q);
  }
  __device__ __forceinline__ void ld12(uint32_t p, uint64_t& lo, uint32_t& hi) const {
    if (in(p, 16)) { LdsAbsSrc().ld12(lds(p), lo, hi); return; }
    uint32_t q; const GlobalSrc f = far(p, q); f.ld12(q, lo, hi);
  }
  __device__ __forceinline__ v4w ld16(uint32_t p) const {
    if (in(p, 20)) return LdsAbsSrc().ld16(lds(p));
    uint32_t q; const GlobalSrc f = far(p, q); return f.ld16(q);
  }
  static constexpr bool kAligned16 = true;
  __device__ __forceinline__ v4w ld16a(uint32_t p) const {      // (p 16-byte aligned -- window positions and wa are)
    if (in(p, 16)) return LdsAbsSrc().ld16a(lds(p));
    uint32_t q; const GlobalSrc f = far(p, q); return f.ld16(q);
  }
  // (copy_bytes' batched pieces: aligned dwords around p -- the same bytes through whichever side holds all of them)
  __device__ __forceinline__ uint32_t ld4a(uint32_t p) const {
    if (in(p & ~3u, 4)) return LdsAbsSrc().ld4a(lds(p));
    uint32_t q; const GlobalSrc f = far(p & ~3u, q); return (uint32_t)f.ld8(q);
  }
  __device__ __forceinline__ void next2(uint32_t p, int j, uint32_t& d1, uint32_t& d2) const {
    if (in((p & ~3u) + (uint32_t)j + 4u, 8)) { LdsAbsSrc().next2(lds(p), j, d1, d2); return; }
    uint32_t q; const GlobalSrc f = far((p & ~3u) + (uint32_t)j + 4u, q);
    const uint64_t x = f.ld8(q);
    d1 = (uint32_t)x; d2 = (uint32_t)(x >> 32);
  }
};

// A RANGE of records that is staged whole, slack included (spec_body.h ranged_tile): no read of a record of the range leaves the
// staged bytes, so the walk reads the window like the kernels of the tiles that fit -- LdsAbsSrc's unchecked aligned reads, no
// in-window test and no global-memory arm behind every head -- while the handlers keep their range-aware forms (kSlide: bitmap
// words accumulated over the ranges, dense lists that count their items as they meet them).  A lane of the fast walk that
// lost its record reads LDS wherever its cursor points (out of range: zero), as it does in those kernels.
struct RangeSrc : LdsAbsSrc {
  static constexpr bool kSlide = true;
  static constexpr bool kMoves = false;
  static constexpr bool sliding = false;
  static constexpr bool lanes = false;
  template <class LaneT>
  __device__ __forceinline__ void refill(LaneT&, uint32_t) const {}
  __device__ __forceinline__ uint32_t advance_to(uint32_t, uint32_t, int) const { return 0; }
  __device__ __forceinline__ void sync(int) const {}
};

// Arrow buffers live in HBM: typed global-address-space accessors keep the compiler from emitting
// flat_* instructions (which tie up both the vector-memory and the LDS counters).
// WIDE = false: the byte offset idx * sizeof(T) is formed in 32 bits, so the store takes the
// uniform-base + 32-bit-lane-offset addressing form (no 64-bit address arithmetic per lane).  The host
// only launches kernels built that way when every buffer of a chunk is smaller than 4 GiB.
template <class T, bool WIDE>
__device__ __forceinline__ void st_global(void* base, uint32_t idx, T v) {
#ifdef RH_V_NOSTORE
  if (reinterpret_cast<uintptr_t>(base) != 1) return;      // (timing-only build: the value is computed, never stored)
#endif
  if (WIDE) {
    reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base))[(uint64_t)idx] = v;
  } else {
    const uint32_t boff = idx * (uint32_t)sizeof(T);
    *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base) + boff) = v;
  }
}
__device__ __forceinline__ void atomic_or_global(void* base, uint64_t idx, uint32_t bits) {
  __hip_atomic_fetch_or(reinterpret_cast<RH_GLOBAL uint32_t*>(reinterpret_cast<uintptr_t>(base)) + idx, bits,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// copy `len` bytes window[sp..] -> d (HBM), any alignment on both sides, with as few store instructions as
// possible: 16-byte stores for strings of 16 bytes and more, and the tail of a string written as the LAST
// 16 (8, 4, 2) bytes ending at its end, overlapping what the first store already wrote, instead of 8+4+2+1-byte
// pieces -- two stores for any length up to 32.  (Unaligned per-lane stores cost ~44 TA cycles per wave
// instruction on MI355X whatever their width -- tools/gmemalign.hip -- so the instruction count is what
// matters.)  All window reads of a round are issued before the first store.  Reads never leave the string.
// A store of T at byte offset `off` from a uniform buffer base.  WIDE = false: the offset stays 32-bit, so the
// instruction takes the scalar-base + 32-bit-lane-offset form (no per-lane 64-bit address arithmetic).
template <class T, bool WIDE>
__device__ __forceinline__ void st_at(void* base, uint64_t off, T v) {
#ifdef RH_V_NOSTORE
  if (reinterpret_cast<uintptr_t>(base) != 1) return;
#endif
  if (WIDE) *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base) + off) = v;
  else *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base) + (uint32_t)off) = v;
}
// the same at a compile-time distance behind `off`: the distance is added in 64 bits, AFTER the 32-bit offset is extended,
// so it lands in the instruction's immediate offset field instead of costing a VALU add per store
template <class T, bool WIDE, int IMM>
__device__ __forceinline__ void st_at_imm(void* base, uint64_t off, T v) {
#ifdef RH_V_NOSTORE
  if (reinterpret_cast<uintptr_t>(base) != 1) return;
#endif
  if (WIDE) *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base) + off + (uint64_t)IMM) = v;
  else *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base) + (uint64_t)(uint32_t)off + (uint64_t)IMM) = v;
}
// byte offsets into a buffer: 64-bit when WIDE, else 32-bit END TO END -- an offset that is summed in 64 bits and
// truncated at the store makes the compiler carry a zero-extended VGPR pair and add the base with a VALU instruction
// (v_lshl_add_u64 + the `off` addressing form) instead of using the scalar-base form (72 VALU instructions of the emit
// walk, k_emit -1.5 %: profiles/r02i_variants_ab.txt)
template <bool WIDE> struct BufOff { typedef uint64_t type; };
template <> struct BufOff<false> { typedef uint32_t type; };

__device__ __forceinline__ uint64_t pick_long(bool mine, uint32_t len, uint32_t& thr);
template <bool WIDE, class Src>
__device__ __forceinline__ void copy_bytes_coop(void* base, typename BufOff<WIDE>::type d, const Src& s, uint32_t sp, uint32_t len, uint64_t big,
                                                uint32_t rank, uint32_t nact);

// (RH_WIDE_SCHEMA, defined by specialize.cpp for a wide schema: copy_bytes is a real function there, called once per string
//  column instead of inlined into every one of them -- the emit kernel of a 200-column schema compiles in 36 s instead of 88 s,
//  for a call per column and wavefront)
#if defined(RH_WIDE_SCHEMA) && !defined(RH_V_INLINECOPY)
#define RH_COPY_FN __attribute__((noinline))
#else
#define RH_COPY_FN __forceinline__
#endif
template <bool WIDE, class Src>
__device__ RH_COPY_FN void copy_bytes(void* base, typename BufOff<WIDE>::type d, const Src s, uint32_t sp, uint32_t len, bool anylong) {      // (the source BY VALUE: a reference into a real function -- RH_WIDE_SCHEMA -- would keep it in scratch memory)
  // A store instruction costs the CU's store path about (width x 64 lanes) / 18 cycles WHATEVER the number of active
  // lanes (tools/storecost.hip): what a column costs that path is its bytes rounded up to pieces, whatever the piece.
#ifndef RH_V_NOBATCH
  // Strings of up to 40 bytes (every string of the wave): ALL window reads of the column's strings are issued before the
  // first store -- the first 8 bytes, the last 8 bytes, and under wave-uniform tests the 8-byte pieces between -- so a string
  // column costs ONE LDS round trip, where head piece, each further piece and the tail each waited for their own
  // (the emit walk is bound by its dependent round trips at 16 waves per CU, not by instruction issue: 16 % fewer VALU
  // instructions moved it 1.5 %, profiles/r04d_*).  Then whole 8-byte pieces at constant offsets + the last 8 bytes,
  // overlapping; strings below 8 bytes by the set bits of their length.
  if (!anylong) {      // (wave-uniform: no string of the wave is longer than 40 bytes)
    const uint32_t sh = sp & 3u;
    uint32_t w0 = s.ld4a(sp), w1, w2, w3, w4, w5, w6, w7, w8, w9, w10;
    // (pieces the wave does not reach are never stored: their registers just need A value, at no cost)
    asm volatile("" : "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7), "=v"(w8), "=v"(w9), "=v"(w10));
    s.next2(sp, 0, w1, w2);
    const bool tail = len > 8u && (len & 7u) != 0;
    const uint32_t tp = sp + (tail ? len - 8u : 0u);
    uint32_t t0 = s.ld4a(tp), t1, t2;
    s.next2(tp, 0, t1, t2);
    if (__any(len >= 16u)) {
      s.next2(sp, 8, w3, w4);
      if (__any(len >= 24u)) {
        s.next2(sp, 16, w5, w6);
        if (__any(len >= 32u)) {
          s.next2(sp, 24, w7, w8);
          if (__any(len >= 40u)) s.next2(sp, 32, w9, w10);
        }
      }
    }
    const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sh), x1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    if (len >= 8u) {
      st_at<u64u, WIDE>(base, d, ((uint64_t)x1 << 32) | x0);
      if (len >= 16u) {
        st_at_imm<u64u, WIDE, 8>(base, d, ((uint64_t)__builtin_amdgcn_alignbyte(w4, w3, sh) << 32) | __builtin_amdgcn_alignbyte(w3, w2, sh));
        if (len >= 24u) {
          st_at_imm<u64u, WIDE, 16>(base, d, ((uint64_t)__builtin_amdgcn_alignbyte(w6, w5, sh) << 32) | __builtin_amdgcn_alignbyte(w5, w4, sh));
          if (len >= 32u) {
            st_at_imm<u64u, WIDE, 24>(base, d, ((uint64_t)__builtin_amdgcn_alignbyte(w8, w7, sh) << 32) | __builtin_amdgcn_alignbyte(w7, w6, sh));
            if (len >= 40u)
              st_at_imm<u64u, WIDE, 32>(base, d, ((uint64_t)__builtin_amdgcn_alignbyte(w10, w9, sh) << 32) | __builtin_amdgcn_alignbyte(w9, w8, sh));
          }
        }
      }
      if (tail) {
        const uint32_t tsh = tp & 3u;
        st_at<u64u, WIDE>(base, d + (len - 8u), ((uint64_t)__builtin_amdgcn_alignbyte(t2, t1, tsh) << 32) | __builtin_amdgcn_alignbyte(t1, t0, tsh));
      }
    } else {
      const uint64_t x = ((uint64_t)x1 << 32) | x0;
      if (len & 4u) st_at<u32u, WIDE>(base, d, (uint32_t)x);
      if (len & 2u) st_at<u16u, WIDE>(base, d + (len & 4u), (uint16_t)(x >> (8 * (len & 4u))));
      if (len & 1u) st_at<uint8_t, WIDE>(base, d + (len & 6u), (uint8_t)(x >> (8 * (len & 6u))));
    }
    return;
  }
#endif
  // longer strings in the wave.  The few longest ones first, by all the lanes that are active here (copy_bytes_coop) ...
  if constexpr (!Src::kSlide) {
    uint32_t thr;
    const uint64_t big = pick_long(true, len, thr);
    if (big) {
      const uint64_t actm = __ballot(true);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(actm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)actm, 0u));
      copy_bytes_coop<WIDE>(base, d, s, sp, len, big, rank, (uint32_t)__popcll(actm));
      if (len >= thr) return;
    }
  }
  // ... the others per lane: 16-byte pieces (two window reads in flight per round), 8..15 bytes as two overlapping
  // 8-byte stores, shorter ones by the set bits of their length
  if (len >= 16) {
    if (Src::kAligned16 && __any(len >= 64u)) {
      // Round 6: the pieces are cut where the SOURCE is 16-byte aligned -- one ds_read_b128 and one (unaligned) 16-byte store
      // per piece, against five aligned dword reads and four v_alignbyte for a piece at any position: the first 16 bytes and
      // the last 16 bytes of the string as before (overlapping the aligned pieces between them).  What a column of strings of
      // a few hundred bytes costs the emit walk fell by a factor of four (the skewed workload, profiles/r06_f_*).
      const v4w xh = s.ld16(sp), xt = s.ld16(sp + len - 16);
      st_at<v4wu, WIDE>(base, d, xh);
      uint32_t j = (16u - (sp & 15u)) & 15u;             // the first aligned source position behind sp
      for (; j + 32 <= len; j += 32) {
        const v4w x0 = s.ld16a(sp + j), x1 = s.ld16a(sp + j + 16);
        st_at<v4wu, WIDE>(base, d + j, x0);
        st_at<v4wu, WIDE>(base, d + j + 16, x1);
      }
      if (j + 16 <= len) st_at<v4wu, WIDE>(base, d + j, s.ld16a(sp + j));
      st_at<v4wu, WIDE>(base, d + len - 16, xt);
      return;
    }
    uint32_t j = 0;
    for (; j + 32 <= len; j += 32) {
      const v4w x0 = s.ld16(sp + j), x1 = s.ld16(sp + j + 16);
      st_at<v4wu, WIDE>(base, d + j, x0);
      st_at<v4wu, WIDE>(base, d + j + 16, x1);
    }
    if (j < len) {   // 1..31 bytes left: [j, j+16) if it fits, then the last 16
      const uint32_t a = j + 16 <= len ? j : len - 16;
      const v4w x0 = s.ld16(sp + a), x1 = s.ld16(sp + len - 16);
      st_at<v4wu, WIDE>(base, d + a, x0);
      if (j + 16 < len) st_at<v4wu, WIDE>(base, d + len - 16, x1);
    }
  } else if (len >= 8) {
    const uint64_t x0 = s.ld8(sp), x1 = s.ld8(sp + len - 8);
    st_at<u64u, WIDE>(base, d, x0);
    if (len > 8) st_at<u64u, WIDE>(base, d + len - 8, x1);
  } else {
    const uint64_t x = s.ld8(sp);   // len <= 7: bytes beyond the string are read (inside the window) but not written
    // one store per set bit of the length (4, 2, 1 bytes): at most three instructions for a wave whose short strings
    // have every length 1..7 (the length classes 4..7 / 2..3 / 1 with overlapping tails took up to five;
    // k_emit -0.3..-1.8 % in two A/B pairs, profiles/r03g_variants_ab.txt)
    if (len & 4u) st_at<u32u, WIDE>(base, d, (uint32_t)x);
    if (len & 2u) st_at<u16u, WIDE>(base, d + (len & 4u), (uint16_t)(x >> (8 * (len & 4u))));
    if (len & 1u) st_at<uint8_t, WIDE>(base, d + (len & 6u), (uint8_t)(x >> (8 * (len & 6u))));
  }
}

template <bool WIDE, class Src>
__device__ __forceinline__ void copy_bytes(void* base, typename BufOff<WIDE>::type d, const Src& s, uint32_t sp, uint32_t len) {
  copy_bytes<WIDE>(base, d, s, sp, len, __any(len > 40u));
}

// Long strings, by the whole wavefront (round 6): a lane that copies an 8 KiB string by itself runs 256 rounds of two 16-byte
// pieces while its 63 neighbours wait; here the strings of `big` (a ballot: few lanes, each with at least kCoopMin bytes) are
// taken one after the other, lane i copying the 16-byte pieces i, i + 64, ... -- 1 KiB per round, the last piece drawn back to
// end with the string (overlapping, like copy_bytes' tails).  Over a SlideSrc (the ranged kernels: a record of its own may hold
// ONE string of megabytes, with one lane active) h_string calls it outside its per-lane region, all 64 lanes taking part; the
// kernels of the tiles that fit call it from copy_bytes' long-string branch, inside the region, with the lanes that are active
// there -- the common path (no string beyond 40 bytes) pays nothing for it.
// pick_long: the smallest threshold (256 B, 1 KiB, 4 KiB ...) that leaves at most kCoopMaxLanes strings to the wavefront, so that
// the per-lane rounds end where many lanes are still busy; 0 = nothing for the wavefront.
constexpr uint32_t kCoopMin = 256;
constexpr uint32_t kCoopMaxLanes = 16;
__device__ __forceinline__ uint64_t pick_long(bool mine, uint32_t len, uint32_t& thr) {
  thr = kCoopMin;
  uint64_t big = __ballot(mine && len >= thr);
  while ((uint32_t)__popcll(big) > kCoopMaxLanes && thr < (1u << 20)) { thr <<= 2; big = __ballot(mine && len >= thr); }
  return (uint32_t)__popcll(big) > kCoopMaxLanes ? 0ull : big;
}
// `rank` of `nact`: this lane's place among the lanes that take part (all 64, or -- inside a per-lane region -- the active ones).
template <bool WIDE, class Src>
__device__ __forceinline__ void copy_bytes_coop(void* base, typename BufOff<WIDE>::type d, const Src& s, uint32_t sp, uint32_t len, uint64_t big,
                                                uint32_t rank, uint32_t nact) {
  while (big) {
    const int j = (int)__builtin_ctzll(big);
    big &= big - 1;
    if constexpr (Src::kSlide) { if (s.sliding) s.sync(j); }      // (the string's own lane holds the window state: SlideSrc::advance_to)
    const uint32_t sp_j = (uint32_t)__builtin_amdgcn_readlane((int)sp, j);
    const uint32_t ln_j = (uint32_t)__builtin_amdgcn_readlane((int)len, j);
    typename BufOff<WIDE>::type d_j;
    if constexpr (WIDE) d_j = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)d >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d, j);
    else d_j = (uint32_t)__builtin_amdgcn_readlane((int)d, j);
    if constexpr (Src::kAligned16) {
      // pieces cut where the source is 16-byte aligned (one ds_read_b128 each); the string's first and last 16 bytes by the
      // first two lanes, overlapping them
      const uint32_t h = (16u - (sp_j & 15u)) & 15u;
      if (rank == 0) st_at<v4wu, WIDE>(base, d_j, s.ld16(sp_j));
      if (rank == (nact > 1 ? 1u : 0u)) st_at<v4wu, WIDE>(base, d_j + ln_j - 16u, s.ld16(sp_j + ln_j - 16u));
      for (uint32_t off = h + rank * 16u; off + 16u <= ln_j; off += nact * 16u) st_at<v4wu, WIDE>(base, d_j + off, s.ld16a(sp_j + off));
    } else {
      for (uint32_t off = rank * 16u; off < ln_j; off += nact * 16u) {
        const uint32_t o = off + 16u <= ln_j ? off : ln_j - 16u;
        const v4w x = s.ld16(sp_j + o);
        st_at<v4wu, WIDE>(base, d_j + o, x);
      }
    }
  }
}

// the low `len` (1..8) bytes of a register -> d, exactly (neighbouring lanes own the neighbouring bytes)
template <bool WIDE>
__device__ __forceinline__ void store_reg8(void* base, typename BufOff<WIDE>::type d, uint64_t x, uint32_t len) {
  if (len & 8u) { st_at<u64u, WIDE>(base, d, x); return; }
  if (len & 4u) st_at<u32u, WIDE>(base, d, (uint32_t)x);
  if (len & 2u) st_at<u16u, WIDE>(base, d + (len & 4u), (uint16_t)(x >> (8 * (len & 4u))));
  if (len & 1u) st_at<uint8_t, WIDE>(base, d + (len & 6u), (uint8_t)(x >> (8 * (len & 6u))));
}

template <class D>
__device__ __forceinline__ void copy_plain(D* d, const uint8_t* s, uint32_t len) {
  for (uint32_t j = 0; j < len; j++) d[j] = s[j];
}

// --------------------------------------------------------------------------
// per-lane walker state
// --------------------------------------------------------------------------
// The bit stacks of a lane.  RH_DEEP (the interpreter always; a specialised kernel when its schema nests deeper than 31 nullable
// records / unions / lists or 8 N-variant unions): 64-bit `live` / `pres` stacks and a 128-bit selector stack -- nesting to 63
// and 16 levels.  apache-avro parses a schema with serde_json, whose recursion limit is 128 JSON levels (an array level costs
// one, a nullable record four): deeper schemas than that cannot reach the reference's decoder either.
#ifdef RH_DEEP
typedef uint64_t stk_t;
typedef unsigned __int128 sel_t;
#else
typedef uint32_t stk_t;
typedef uint64_t sel_t;
#endif
struct Lane {
  uint32_t cur, end;   // byte cursor / record end, relative to the window base (cur <= end always)
  uint32_t err;        // ErrCode, 0 = ok
  int64_t edetail;
  bool live, pres;     // an errored lane is dead: live = pres = false and its saved bits are cleared
  bool redo;           // fast walk only: this record left the fast wire forms and must be walked carefully
  uint64_t la;         // look-ahead: the window bytes at `cur`, left behind by the head in front (read_head LA bit 1)
  stk_t pstk;          // saved `pres` bits   (nullable record / union / list)
  stk_t lstk;          // saved `live` bits   (list)
  sel_t sstk;          // saved union selectors, 8 bits each
};

// First error of a record (the reference's `?` at fast_decode.rs:827): remember it and kill the lane so
// that every later handler sees it as not-live (nothing is emitted or counted for it any more).
__device__ __forceinline__ void fail(Lane& L, uint32_t code, int64_t detail = 0) {
  L.err = code;
  L.edetail = detail;
  L.live = false;
  L.pres = false;
  L.pstk = 0;
  L.lstk = 0;
}

// Two forms of every walk.  CAREFUL: anomalies (malformed input, and wire forms outside the single-read
// fast path) are resolved on the spot, behind wave-uniform branches, with the reference's exact error order.
// FAST (CAREFUL = false): no such branch exists at all -- a lane that meets ANY anomaly just marks `redo`
// and dies, fully predicated; whoever ran the fast walk re-runs the careful one for that wave / tile.  The
// size kernel finds out which tiles need it (almost none do) and tells the emit kernel (tileflag bit 1).
template <bool CAREFUL>
__device__ __forceinline__ void reject(Lane& L, bool cond, uint32_t code, int64_t detail = 0) {
  if (CAREFUL) {
    if (__any(cond)) {
      if (cond) fail(L, code, detail);
    }
  } else {
    L.redo = L.redo || cond;
    L.live = L.live && !cond;
    L.pres = L.pres && !cond;
    L.pstk = cond ? (stk_t)0 : L.pstk;
    L.lstk = cond ? (stk_t)0 : L.lstk;
  }
}

// The emit kernel only takes the fast walk for tiles whose size pass (same predicates, same bytes) met no anomaly in
// any lane (tileflag bit 1 clear), so in <EMIT, !CAREFUL> the anomaly predicates are dead weight: reject() and
// everything that only feeds it drop out (measured on MI355X: k_emit 0.950 -> 0.918 ms, profiles/r02a_variants_ab.txt).
// (Ctx::kSkip: the same holds for the size-like sub-walks the emit kernel runs inside such a tile -- SkipCtx below.)
#define RH_TRUST ((EMIT && !CAREFUL) || Ctx::kSkip)
#define RH_REJECT(L, ...) do { if constexpr (!RH_TRUST) reject<CAREFUL>(L, __VA_ARGS__); } while (0)
// The fast walk's common anomalies (a head outside the single-read forms, a negative length, a bad boolean) only MARK
// the lane: the whole wave is walked again carefully anyway, so the lane goes on with whatever it decoded -- its results
// are never used -- and leaves at the next list iteration (h_list_next), which is what bounds its work.  No live / pres /
// stack upkeep per head: k_size -2 % (profiles/r03bf_soft_rejects_ab.txt).  Anomalies that guard a memory access or a
// loop (enum / union index range, the N4 leaves, list block headers) still take the lane out on the spot (RH_REJECT).
#define RH_REJECT_SOFT(L, cond, ...) do { if constexpr (!RH_TRUST) { if constexpr (CAREFUL) reject<true>(L, cond, __VA_ARGS__); else L.redo = L.redo || (cond); } } while (0)

// --------------------------------------------------------------------------
// primitive readers (fast_decode.rs:845-922)
//
// Every field starts with ONE unaligned 8-byte read at the cursor, decoded branch-free for the
// common wire forms (single-byte union branch, varints of <= 4 / <= 8 bytes inside the record).
// Anything else -- longer or non-canonical varints, a varint running past the record end, bad
// branch bytes -- takes the byte-at-a-time path below, which follows the reference's exact error
// order.  That path sits behind a wave-uniform `__any` so the common case never touches EXEC.
// --------------------------------------------------------------------------
template <class Src>
__device__ __forceinline__ uint32_t rd_varint_slow(const Src& src, uint32_t& cur, uint32_t end, int64_t& out) {
  uint64_t r = 0;
  uint32_t shift = 0;
  for (;;) {
    if (cur >= end) return E_EOB;
    uint32_t b = src.ld1(cur++);
    r |= (uint64_t)(b & 0x7F) << shift;
    if ((b & 0x80) == 0) break;
    shift += 7;
    if (shift >= 64) return E_VARINT;
  }
  out = (int64_t)(r >> 1) ^ -(int64_t)(r & 1);
  return E_OK;
}

// The single-read wire forms of the fast walk (round 6: sized for what production data carries, not for the benchmark
// generator -- VERDICT round 5, item 2; the reference's read_zigzag_long, fast_decode.rs:854-869, costs the same for any length):
//   int                       <= 5 bytes  (every i32; longer encodings of an int are legal and take the careful form)
//   long                      <= 10 bytes (every i64, behind a branch byte too: timestamps in microseconds, snowflake ids)
//   string / bytes length     <= 3 bytes  (below 2^20: 1 MiB)
//   array / map block count   <= 4 bytes  (below 2^27 items)
//   union / enum index        <= 2 bytes
// Anything else -- padded encodings beyond these widths, an 11th byte, a varint running past the record -- is an anomaly.

// zig-zag varint of <= 5 bytes at bit 0 of y (>= 5 valid bytes), decoded and truncated the way the reference's `as i32`
// truncates (fast_decode.rs:424,430): bits 33 and 34 of a 5-byte encoding never reach the value.
__device__ __forceinline__ bool varint35(uint64_t y, uint32_t avail, uint32_t& v32, uint32_t& n) {
  uint32_t lo = (uint32_t)y;
  const uint32_t b4 = (uint32_t)(y >> 32) & 0xFFu;
  const uint32_t t = ~lo & 0x80808080u;            // bit 7 of every byte WITHOUT a continuation flag
  const bool ext = t == 0;                         // four continuation bytes: the fifth byte ends the varint
  n = ext ? 5u : (uint32_t)(__ffs((int)t) + 7) >> 3;
  lo &= t ^ (t - 1);                               // keep bytes 0..n-1 (all four when ext: 0 ^ 0xFFFFFFFF)
  lo = ((lo & 0x7F007F00u) >> 1) | (lo & 0x007F007Fu);
  uint32_t raw = ((lo & 0x3FFF0000u) >> 2) | (lo & 0x00003FFFu);
  const uint32_t e4 = ext ? b4 : 0u;
  raw |= e4 << 28;                                 // raw bits 28..31
  v32 = ((raw >> 1) | ((e4 & 0x10u) << 27)) ^ (0u - (raw & 1u));      // raw bit 32 -> value bit 31
  return (e4 & 0x80u) == 0 && n <= avail;
}

// (A/B knobs, scripts/gpu_ab.sh: RH_V_INT28 = the 4-byte int form of rounds 1-5, RH_V_LEN16 = lengths / counts in the 2-byte form)
__device__ __forceinline__ bool varint32(uint32_t y, uint32_t avail, uint32_t& raw, uint32_t& n) {
  const uint32_t t = ~y & 0x80808080u;
  n = (uint32_t)(__ffs((int)t) + 7) >> 3;
  y &= t ^ (t - 1);
  y = ((y & 0x7F007F00u) >> 1) | (y & 0x007F007Fu);
  raw = ((y & 0x3FFF0000u) >> 2) | (y & 0x00003FFFu);
  return t != 0 && n <= avail;
}

// length / block count: raw (pre-zigzag) value and byte length of a varint of <= 3 bytes at bit 0 of y
__device__ __forceinline__ bool varint24(uint32_t y, uint32_t avail, uint32_t& raw, uint32_t& n) {
#ifdef RH_V_LEN16
  const uint32_t m = (uint32_t)((int32_t)(y << 24) >> 31);
  raw = ((((y >> 8) & 0x7Fu) & m) << 7) | (y & 0x7Fu);
  n = 1u - m;
  return (y & 0x8080u) != 0x8080u && n <= avail;
#endif
  const uint32_t m0 = (uint32_t)((int32_t)(y << 24) >> 31);          // all ones when byte 0 carries a continuation flag
  const uint32_t m1 = (uint32_t)((int32_t)(y << 16) >> 31) & m0;     // ... and byte 1 too
  raw = (y & 0x7Fu) | ((((y >> 8) & 0x7Fu) & m0) << 7) | ((((y >> 16) & 0x7Fu) & m1) << 14);
  n = 1u - m0 - m1;                                                  // 1, 2 or 3
  return (y & 0x808080u) != 0x808080u && n <= avail;
}

// union / enum index: the same for <= 2 bytes
// (k_size 0.375 -> 0.358 ms on the full schema against the 4-byte form, profiles/r02d_variants_ab.txt; the size pass is VALU-issue bound.)
constexpr bool kNarrow = true;
__device__ __forceinline__ bool varint16(uint32_t y, uint32_t avail, uint32_t& raw, uint32_t& n) {
  const uint32_t m = (uint32_t)((int32_t)(y << 24) >> 31);     // all ones when byte 0 carries a continuation flag
  raw = ((((y >> 8) & 0x7Fu) & m) << 7) | (y & 0x7Fu);
  n = 1u - m;                                                   // 1 or 2
  return (y & 0x8080u) != 0x8080u && n <= avail;
}

// zig-zag varint of <= 8 bytes at bit 0 of x, full 64-bit value (the careful walk's first try: rd_varint)
__device__ __forceinline__ bool varint64(uint64_t x, uint32_t nx, uint32_t avail, int64_t& out, uint32_t& n) {
  const uint64_t t = ~x & 0x8080808080808080ull;
  if (t == 0) return false;
  n = ((uint32_t)__builtin_ctzll(t) >> 3) + 1;
  if (n > nx || n > avail) return false;
  x &= t ^ (t - 1);
  x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
  x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
  x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
  out = (int64_t)(x >> 1) ^ -(int64_t)(x & 1);
  return true;
}

// zig-zag varint of <= 10 bytes, branch-free: bytes 0..7 in lo, bytes 8..9 in the low half of hi.  The tenth byte's payload
// is shifted by 63: only its bit 0 reaches the value, as in the reference (fast_decode.rs:859: `<< shift` drops the rest).
__device__ __forceinline__ bool varint64x(uint64_t lo, uint32_t hi, uint32_t avail, int64_t& out, uint32_t& n) {
  const uint64_t t = ~lo & 0x8080808080808080ull;
  const bool ext = t == 0;                          // eight continuation bytes
  const uint32_t th = ~hi & 0x8080u;
  const uint32_t nl = ((uint32_t)__builtin_ctzll(t | (1ull << 63)) >> 3) + 1;      // 1..8
  const uint32_t nh = ((uint32_t)__builtin_ctz(th | 0x8000u) >> 3) + 1;            // 1..2
  n = ext ? 8u + nh : nl;
  uint64_t x = lo & (t ^ (t - 1));                  // bytes 0..n-1 (all eight when ext)
  x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
  x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
  x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
  const uint64_t e8 = ext ? (uint64_t)(hi & 0x7Fu) : 0ull;
  const uint64_t e9 = (ext && nh == 2u) ? (uint64_t)((hi >> 8) & 1u) : 0ull;
  x |= (e8 << 56) | (e9 << 63);
  out = (int64_t)(x >> 1) ^ -(int64_t)(x & 1);
  return (!ext || th != 0) && n <= avail;
}

template <class Src>
__device__ __forceinline__ uint32_t rd_varint(const Src& src, uint32_t& cur, uint32_t end, int64_t& out) {
  uint32_t n;
  if (varint64(src.ld8(cur), 8, end - cur, out, n)) { cur += n; return E_OK; }
  return rd_varint_slow(src, cur, end, out);
}

// exact form of [union_branch (585-593)] [value varint (854-869)] at L.cur; advances L.cur, fails the lane
template <class Src>
__device__ __forceinline__ bool read_head_slow(const Src& src, Lane& L, bool nullable, bool null_first, bool want_varint, int64_t& v) {
  bool isval = true;
  if (nullable) {
    int64_t idx = 0;
    const uint32_t e = rd_varint_slow(src, L.cur, L.end, idx);
    if (e) { fail(L, e); return false; }
    if (idx != 0 && idx != 1) { fail(L, E_BRANCH, idx); return false; }
    isval = (idx == 0) ? !null_first : null_first;
  }
  if (isval && want_varint) {
    const uint32_t e = rd_varint(src, L.cur, L.end, v);
    if (e) { fail(L, e); return false; }
  }
  return isval;
}

// Head of a field for the lanes with `dec`: an optional single-byte null-union branch and an optional
// varint (`wide`: may need more than 28 bits).  Returns isval (false for lanes without `dec`); v is the
// varint when isval && want_varint.  L.cur moves past what was read.
// Head fusion (LA, set per op by the generator -- specialize.cpp `la`): heads that are STATICALLY adjacent in the datum -- a
// nullable record's branch byte and its first field's head, a boolean and the union index behind it, a union index and
// the head of whichever variant it selects -- are decoded out of ONE window read.  Bit 1: this head leaves the bytes
// behind what it consumed in L.la (every lane reads at its own cursor and shifts by what IT consumed, so L.la is valid
// for every lane, decoding or not); (LA >> 2) = bytes the first head of a chain reads (4 / 8).  Bit 0: this head takes
// its bytes from L.la and issues no read.  Chains never cross a string body, a loop boundary or 8 bytes, and exist only
// in the fast walks (a careful walk may re-read a head byte by byte and would leave L.la stale).
template <bool CAREFUL, bool TRUST = false, int LA = 0, class Src>
__device__ __forceinline__ bool read_head(const Src& src, Lane& L, bool dec, bool nullable, bool null_first, bool want_varint,
                                         bool wide, int64_t& v, int small = 0) {
  if (!nullable && !want_varint) return dec;
  static_assert(LA == 0 || !CAREFUL, "head fusion is for the fast walks");
  // `small`: 1 = a union / enum index (varint16), 2 = a length / block count (varint24); 0 = a value (int: varint35, long: varint64x)
  const bool narrow = kNarrow && small != 0 && !wide;
  uint64_t x;
  uint32_t xh = 0;                                    // bytes 8..11 behind the cursor (a long only)
  if constexpr ((LA & 1) != 0) x = L.la;
  else if constexpr ((LA & 2) != 0 && (LA >> 2) == 8) x = src.ld8(L.cur);
  else if constexpr ((LA & 2) != 0) x = (uint64_t)src.ld4(L.cur);
  else if (want_varint && wide) src.ld12(L.cur, x, xh);
#ifdef RH_V_INT28
  else x = (narrow || (kNarrow && !want_varint)) ? (uint64_t)src.ld4(L.cur) : src.ld5(L.cur);
#else
  else x = (narrow || (kNarrow && !want_varint)) ? (uint64_t)src.ld4(L.cur) : nullable ? src.ld8(L.cur) : src.ld5(L.cur);
#endif
  // The fast size walk (neither careful nor trusted) checks a record's bounds ONCE, at its end (spec_size: cursor past
  // the record's end -> the wave is walked again, carefully): a cursor only ever moves forward, so a read that runs past
  // the end leaves it past the end for good, and what such a lane decodes meanwhile is never used.  No compare against
  // the bytes left per head (k_size -6...-9 %, profiles/r03bb_deferred_bounds_ab.txt).  List block headers keep theirs
  // (h_list_next): they are what bounds the work of a lane that has lost its record.
  const uint32_t avail = (!CAREFUL && !TRUST) ? 0x7FFFFFFFu : L.end - L.cur;
  uint32_t skip = 0;
  bool okb = true, isval = dec;
  uint64_t y = x;
  if (nullable) {
    const uint32_t b0 = (uint32_t)x & 0xFFu;
    okb = avail != 0 && (b0 & 0xFDu) == 0;          // branch 0 or 1 as a single byte (0x00 / 0x02)
    isval = dec && (b0 == (null_first ? 2u : 0u));
    skip = 1;
    y = x >> 8;
  }
  uint32_t n = 0;
  bool okv = true;
  if (want_varint) {
    const uint32_t av = avail - skip;               // wraps only when okb is false
    if (wide) {
      okv = varint64x(nullable ? (y | ((uint64_t)xh << 56)) : x, nullable ? xh >> 8 : xh, av, v, n);
    } else if (narrow) {
      uint32_t raw;
      okv = small == 2 ? varint24((uint32_t)y, av, raw, n) : varint16((uint32_t)y, av, raw, n);
      if (!CAREFUL) {
        // a length / index / count is never negative in a well-formed record: on the fast walk the sign bit of the
        // zig-zag form is one more anomaly (the careful walk raises the reference's error for it) and the value is
        // just raw >> 1 -- no zig-zag decode, no sign tests downstream (~3 VALU per varint, ~17 of them per record)
        okv = okv && (raw & 1u) == 0;
        v = (int64_t)(raw >> 1);              // (k_size -2.5 %, k_emit -2.3 %: profiles/r03y_variants_ab.txt)
      } else {
        v = (int64_t)(int32_t)((raw >> 1) ^ (0u - (raw & 1u)));
      }
    } else {
#ifdef RH_V_INT28
      uint32_t raw;
      okv = varint32((uint32_t)y, av, raw, n);
      v = (int64_t)(int32_t)((raw >> 1) ^ (0u - (raw & 1u)));
#else
      uint32_t v32;
      okv = varint35(y, av, v32, n);
      v = (int64_t)(int32_t)v32;
#endif
    }
  }
  const bool slow = dec && (!okb || (isval && !okv));
  uint32_t adv = skip + ((isval && want_varint) ? n : 0u);
  if constexpr ((LA & 2) != 0) L.la = x >> (8u * (dec ? adv : 0u));
  if (TRUST) {               // the size pass saw okb && okv on every lane of this tile: `slow` is dead
    L.cur += dec ? adv : 0u;
    return isval;
  }
  if (CAREFUL) {
    if (__any(slow)) {
      if (slow) {
        isval = read_head_slow(src, L, nullable, null_first, want_varint, v);
        adv = 0;
      }
    }
  } else {
    L.redo = L.redo || slow;       // (marked, not taken out: see RH_REJECT_SOFT)
  }
  L.cur += dec ? adv : 0u;
  return isval;
}

// --------------------------------------------------------------------------
// shared output helpers.  Ctx provides:
//   uint32_t& counter(int id)   per-lane counter (child rows / string bytes), block-local after the scan
//   uint32_t& remaining(int d)  items left in the current block of list depth d
//   void* buf(int id)           this chunk's Arrow buffer `id`
//   uint32_t gbase(int id)      chunk-relative base of this workgroup for counter id
//   void add_nulls_wave(int node, uint32_t n)   n is wave-uniform (called by every lane)
//   void add_nulls_lane(int node)               one null row of this lane (child domains)
//   void set_bit(int buf, int dom, uint32_t row) set one bit of a CHILD-domain bitmap (rows there do not line up with lanes)
//   void put_word0(int buf, uint64_t m)          this wavefront's 64 bits of a DOMAIN-0 bitmap (rows == lanes: a ballot)
//   lrow, lane, wave_live, sym_off, sym_data
//   kWaveCtr / wave_total(id, len) / wave_offset(id, len)   wide schemas (program.h F_WAVE_CTR): the byte counter of a domain-0
//       column has no per-lane state -- the size walk adds the wavefront's sum of `len` to the tile's total of counter id, the
//       emit walk returns this lane's chunk-relative byte offset (the wavefront's running base + the exclusive scan of `len`)
// --------------------------------------------------------------------------
template <class Ctx>
__device__ __forceinline__ uint32_t row_of(const Ctx& c, int dom) {
  return dom == 0 ? c.lrow : c.gbase(dom - 1) + c.counter(dom - 1);
}

// View of a context for the counters-only (EMIT = false) walk of bytes the size pass has already cleared: every
// anomaly predicate drops out (RH_TRUST).  Used by the emit kernel's item-dense list handling (spec_body.h dense_list)
// to find item boundaries and item sizes without emitting anything.
template <class C>
struct SkipCtx {
  static constexpr bool kSkip = true;
  static constexpr bool kWide = C::kWide;
  static constexpr bool kEnumImm = C::kEnumImm;
  static constexpr bool kWaveCtr = false;             // (the bodies of lists hold per-lane counters only)
  __device__ __forceinline__ void wave_total(int, uint32_t) const {}
  __device__ __forceinline__ uint32_t wave_offset(int, uint32_t) const { return 0; }
  static __device__ __forceinline__ bool enum_sym(int b, uint32_t v, uint32_t& len, uint64_t& bits) { return C::enum_sym(b, v, len, bits); }
  const C& base;
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  uint32_t lrow, lane;
  bool wave_live;
  __device__ __forceinline__ explicit SkipCtx(const C& b)
      : base(b), sym_off(b.sym_off), sym_data(b.sym_data), lrow(b.lrow), lane(b.lane), wave_live(b.wave_live) {}
  __device__ __forceinline__ uint32_t& counter(int id) const { return base.counter(id); }
  __device__ __forceinline__ uint32_t& remaining(int d) const { return base.remaining(d); }
  // (the EMIT side of the handlers is dead code under EMIT = false, but it has to compile)
  __device__ __forceinline__ void* buf(int id) const { return base.buf(id); }
  __device__ __forceinline__ uint32_t gbase(int id) const { return base.gbase(id); }
  template <bool ACC> __device__ __forceinline__ void add_nulls_wave(int node, uint32_t n) const { base.template add_nulls_wave<ACC>(node, n); }
  __device__ __forceinline__ void add_nulls_lane(int node) const { base.add_nulls_lane(node); }
  template <bool ACC> __device__ __forceinline__ void put_word0(int b, uint64_t m) const { base.template put_word0<ACC>(b, m); }
  __device__ __forceinline__ void set_bit(int b, int dom, uint32_t row) const { base.set_bit(b, dom, row); }
};

// validity bit + null count of one row (the buffer exists iff F_CAN_NULL)
// ACC (Src::kSlide): a wavefront may walk its records in several ranges (spec_body.h ranged_tile): its bitmap word and its
// null count of a domain-0 field are then accumulated (OR / add onto zeroed words) instead of stored.
template <bool EMIT, bool ACC = false, class Ctx>
__device__ __forceinline__ void put_validity(const Ctx& c, const Op& op, bool act, bool valid, uint32_t row) {
  if (!EMIT) return;
  if (!(op.flags & F_CAN_NULL)) return;
  if (op.dom == 0) {   // rows == lanes: one ballot, one 64-bit store per wavefront
    const uint64_t m = __ballot(valid);
    const uint64_t nm = __ballot(act && !valid);
    c.template put_word0<ACC>(op.buf0, m);
    c.template add_nulls_wave<ACC>(op.node, (uint32_t)__popcll(nm));
  } else if (act) {
    if (valid) c.set_bit(op.buf0, op.dom, row);
    else c.add_nulls_lane(op.node);
  }
}

// --------------------------------------------------------------------------
// field handlers
// --------------------------------------------------------------------------
// int/long/float/double/boolean/date/timestamp leaf, optionally Nullable* (424-432, 434-473)
template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_fixed(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  const bool dec = act && L.pres;
  const bool is_int = op.a == FK_I32 || op.a == FK_I64;
  int64_t v = 0;
  void* const pf1 = (EMIT && op.a != FK_BOOL) ? c.buf(op.buf1) : nullptr;     // requested ahead of the head: see h_string
  const bool isval = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, (op.flags & F_NULLABLE) != 0, (op.flags & F_NULL_FIRST) != 0, is_int,
                                        op.a == FK_I64, v);
  uint64_t bits;
  bool valid;
  if (is_int) {
    valid = isval && L.live;
    bits = op.a == FK_I32 ? (uint64_t)(uint32_t)(int32_t)v : (uint64_t)v;   // `as i32` truncates (424,430)
  } else {
    // (head fusion, read_head: a head-less leaf -- boolean / float without a null union -- may open or continue a chain)
    constexpr int la = CAREFUL ? 0 : LA;
    uint64_t x;
    if constexpr ((la & 1) != 0) x = L.la;
    else if constexpr ((la & 2) != 0 && (la >> 2) == 8) x = src.ld8(L.cur);
    else if constexpr ((la & 2) != 0) x = (uint64_t)src.ld4(L.cur);
    else x = op.a == FK_F64 ? src.ld8(L.cur) : op.a == FK_F32 ? src.ld5(L.cur) : (uint64_t)src.ld1(L.cur);
    const uint32_t avail = L.end - L.cur;
    const uint32_t need = op.a == FK_F32 ? 4u : op.a == FK_F64 ? 8u : 1u;
    const bool want = isval && L.live;
    const bool eob = (!CAREFUL && !RH_TRUST) ? false : (want && avail < need);      // (fast size walk: see read_head)
    bits = op.a == FK_F32 ? (uint64_t)(uint32_t)x : op.a == FK_F64 ? x : (x & 0xFFu);
    const bool badb = want && !eob && op.a == FK_BOOL && bits > 1;   // read_bool, 893-900
    RH_REJECT_SOFT(L, eob, op.a == FK_F32 ? E_EOB_F32 : op.a == FK_F64 ? E_EOB_F64 : E_EOB);
    RH_REJECT_SOFT(L, badb, E_BOOL, (int64_t)bits);
    valid = want && L.live;
    L.cur += valid ? need : 0u;
    if constexpr ((la & 2) != 0) L.la = x >> (8u * (valid ? need : 0u));
  }
  if (!valid) bits = 0;   // zero under nulls (arrow-rs append_null)
  uint32_t row = 0;
  if (EMIT) {
    row = row_of(c, op.dom);
    if (op.a == FK_BOOL) {
      if (op.dom == 0) {
        const uint64_t m = __ballot(bits != 0);
        c.template put_word0<Src::kSlide>(op.buf1, m);
      } else if (act && bits) {
        c.set_bit(op.buf1, op.dom, row);
      }
    } else if (act) {
      if (op.a == FK_I32 || op.a == FK_F32) st_global<uint32_t, Ctx::kWide>(pf1, row, (uint32_t)bits);
      else st_global<uint64_t, Ctx::kWide>(pf1, row, bits);
    }
  }
  put_validity<EMIT, Src::kSlide>(c, op, act, valid, row);
}

// string leaf / map key (429, 454-457, 752, read_string 902-922) and enum -> symbol text (570-578)
template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_string(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  const bool dec = act && L.pres;
  int64_t v = 0;
  // The buffer addresses (scalar loads through the constant cache) are requested before the head is read, so their
  // latency runs beside the LDS read instead of in front of each store (k_emit -1.2 %, profiles/r03at_variants_ab.txt)
  void* const pb1 = EMIT ? c.buf(op.buf1) : nullptr;
  void* const pb2 = EMIT ? c.buf(op.buf2) : nullptr;
  const bool isval = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, (op.flags & F_NULLABLE) != 0, (op.flags & F_NULL_FIRST) != 0, true, false, v, op.code == OP_STRING ? 2 : 1);
  const bool want = isval && L.live;
  uint32_t len = 0, spos = 0;
  bool sym_imm = false;
  uint64_t sym_bits = 0;
  if (op.code == OP_STRING) {
    // the fast walk only ever sees lengths from the 28-bit single-read decode: 32-bit compares are enough there
    const bool neg = want && (CAREFUL ? v < 0 : (int32_t)v < 0);
    // (the fast size walk checks a record's bounds once, at its end: see read_head)
    const bool eob = (!CAREFUL && !RH_TRUST) ? false : (want && !neg && (CAREFUL ? (uint64_t)v > (uint64_t)(L.end - L.cur) : (uint32_t)v > L.end - L.cur));
    RH_REJECT_SOFT(L, neg, E_NEGLEN);
    RH_REJECT_SOFT(L, eob, E_EOB_STR);
    len = (want && L.live) ? (uint32_t)v : 0u;
    spos = L.cur;
    L.cur += len;
  } else {
    const bool oor = want && (CAREFUL ? (uint64_t)v >= (uint64_t)op.c : (uint32_t)v >= (uint32_t)op.c);
    RH_REJECT(L, oor, E_ENUM, v);
    // few, short symbols: (length, bytes) selected from immediates by the index.  Only where the length is all that is
    // needed (the size pass: k_size -1.2 %); the emit walk is 2.2 % SLOWER with the immediates than with its two trips
    // to memory (three A/B pairs, profiles/r03au_enum_immediates_ab.txt)
    if constexpr (Ctx::kEnumImm && !EMIT) {
      uint32_t l0 = 0;
      sym_imm = c.enum_sym(op.b, (uint32_t)v, l0, sym_bits);
      if (sym_imm) len = (want && L.live) ? l0 : 0u;
    }
    if (!sym_imm && want && L.live) {
      spos = c.sym_off[op.b + (int32_t)v];
      len = c.sym_off[op.b + (int32_t)v + 1] - spos;
    }
  }
  const bool valid = want && L.live;
  // the byte counter: per lane (block-local after the emit kernel's scan), or -- wide schemas, domain 0 -- a wave counter
  bool wctr = false;
  if constexpr (Ctx::kWaveCtr) wctr = (op.flags & F_WAVE_CTR) != 0;
  uint32_t o = 0;
  if (wctr) {
    if constexpr (EMIT) o = c.wave_offset(op.a, len);
    else c.wave_total(op.a, len);
  } else {
    o = c.counter(op.a);
  }
  uint32_t row = 0;
  if (EMIT) {
    row = row_of(c, op.dom);
    // wave-uniform, outside the per-lane region: is any string of this column longer than copy_bytes' batched form takes, and
    // which few of them are long enough for the whole wavefront to copy (copy_bytes_coop)
    // (over a SlideSrc the long strings are picked here, outside the per-lane region: all 64 lanes copy them -- copy_bytes_coop)
    bool anylong = false, coop = false;
    uint64_t big = 0;
    if constexpr (Src::kSlide) {
      if (op.code == OP_STRING) {
        anylong = __any(act && len > 40u);
        if (anylong && !src.lanes) {      // (lane windows: a long string is copied by its own lane)
          uint32_t thr;
          big = pick_long(act, len, thr);
          coop = big != 0 && len >= thr;
        }
      }
    }
    const uint32_t gb = wctr ? 0u : c.gbase(op.a);                   // (a wave counter's offset is chunk-relative already)
    if (act) {
      st_global<uint32_t, Ctx::kWide>(pb1, row + 1, gb + o + len);   // offsets repeat under nulls
      if (len && !coop) {
        // String bytes go straight to HBM with per-lane 8-byte stores at any alignment: neighbouring lanes
        // own neighbouring rows, so one wave store covers one contiguous span of the column.  (Staging the
        // column in LDS and flushing it with aligned 16-byte stores was measured slower: the extra LDS halves
        // the workgroups per CU, and this walk is latency-bound -- DESIGN.md, "string bytes".)
        if (op.code == OP_STRING) {
          if constexpr (Src::kSlide) copy_bytes<Ctx::kWide>(pb2, (typename BufOff<Ctx::kWide>::type)gb + o, src, spos, len, anylong);
          else copy_bytes<Ctx::kWide>(pb2, (typename BufOff<Ctx::kWide>::type)gb + o, src, spos, len);
        } else {
          if (sym_imm) {
            store_reg8<Ctx::kWide>(pb2, (typename BufOff<Ctx::kWide>::type)gb + o, sym_bits, len);
          } else {
            RH_GLOBAL uint8_t* d = reinterpret_cast<RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(pb2)) + gb + o;
            copy_plain(d, c.sym_data + spos, len);
          }
        }
      }
    }
    if constexpr (Src::kSlide) {
      if (big) {      // by the lanes that are active here: all 64, unless the compiler runs an enclosing block loop with its live lanes only
        const uint64_t ex = __ballot(true);
        copy_bytes_coop<Ctx::kWide>(pb2, (typename BufOff<Ctx::kWide>::type)gb + o, src, spos, len, big,
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u)), (uint32_t)__popcll(ex));
      }
    }
  }
  if (!wctr) c.counter(op.a) = o + len;   // len == 0 for every lane that does not carry a value
  put_validity<EMIT, Src::kSlide>(c, op, act, valid, row);
}

// --------------------------------------------------------------------------
// fixed / decimal / uuid leaves.  The reference translates these schemas to Arrow
// (schema_translate.rs:133-137: fixed -> FixedSizeBinary(n), decimal -> Decimal128(p, s), uuid -> FixedSizeBinary(16))
// but never decodes them on its direct path (fast_decode.rs:59), and its Value-tree fallback has no builder for
// them (complex.rs:414-431): the wire forms below are the Avro 1.11 specification's.  Null slots are zero-filled
// like every arrow-rs fixed-width builder does.  Not a hot path: byte loops, no fast wire form.
// --------------------------------------------------------------------------
__device__ __forceinline__ int hex_nibble(uint32_t ch) {
  if (ch >= '0' && ch <= '9') return (int)(ch - '0');
  ch |= 0x20u;
  if (ch >= 'a' && ch <= 'f') return (int)(ch - 'a' + 10);
  return -1;
}

template <bool WIDE>
__device__ __forceinline__ void fill_zero(void* base, uint64_t off, uint32_t n) {
  uint32_t j = 0;
  for (; j + 8 <= n; j += 8) st_at<u64u, WIDE>(base, off + j, 0ull);
  for (; j < n; j++) st_at<uint8_t, WIDE>(base, off + j, (uint8_t)0);
}

template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_bin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  const bool dec = act && L.pres;
  const bool has_len = op.a == BN_DEC_BYTES || op.a == BN_UUID_STR;
  const uint32_t W = (uint32_t)op.c;
  int64_t v = 0;
  const bool isval = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, (op.flags & F_NULLABLE) != 0, (op.flags & F_NULL_FIRST) != 0, has_len, false, v, 2);
  const bool want = isval && L.live;
  uint32_t len = (uint32_t)op.b;
  if (has_len) {
    const bool neg = want && (CAREFUL ? v < 0 : (int32_t)v < 0);
    // bytes left in the record, 0 for a cursor that is already past its end: the fast size walk checks a record's bounds
    // once, at its end (read_head), so its cursor may have run past L.end after an over-long string, and the unsigned
    // difference would wrap and let the guards below pass -- these guards are what keeps the byte loops inside the window
    const uint32_t left = (int32_t)(L.end - L.cur) < 0 && !CAREFUL ? 0u : L.end - L.cur;
    const bool eob = want && !neg && (CAREFUL ? (uint64_t)v > (uint64_t)left : (uint32_t)v > left);
    RH_REJECT(L, neg, E_NEGLEN);
    RH_REJECT(L, eob, E_EOB_STR);
    len = (want && L.live) ? (uint32_t)v : 0u;
    if (op.a == BN_DEC_BYTES) RH_REJECT(L, want && L.live && len > 16u, E_DECIMAL, (int64_t)len);
    if (op.a == BN_UUID_STR) RH_REJECT(L, want && L.live && len != 36u && len != 32u, E_UUID);
  } else {
    const uint32_t left = (int32_t)(L.end - L.cur) < 0 && !CAREFUL ? 0u : L.end - L.cur;      // (see above)
    RH_REJECT(L, want && left < len, E_EOB_FIXED);
  }
  bool valid = want && L.live;
  const uint32_t spos = L.cur;
  // value bits: 16 bytes for decimal / uuid (computed in both passes for uuid: malformed text is an error)
  uint64_t lo = 0, hi = 0;
  if (op.a == BN_DEC_BYTES || op.a == BN_DEC_FIXED) {
    if (EMIT && valid) {
      const bool negv = len > 0 && (src.ld1(spos) & 0x80u);
      lo = negv ? ~0ull : 0ull; hi = lo;                       // sign extension of a value shorter than 16 bytes
      for (uint32_t j = 0; j < len; j++) {
        const uint64_t b = src.ld1(spos + j);
        hi = (hi << 8) | (lo >> 56);
        lo = (lo << 8) | b;
      }
    }
  } else if (op.a == BN_DURATION) {
    // months / days / milliseconds, three little-endian u32.  Arrow's Duration(ms) (the reference's mapping,
    // schema_translate.rs:143) is ONE count of milliseconds: days x 86 400 000 + milliseconds; a months component has no
    // value in it, so a duration that carries one is a decode error (found by the size pass, like every other)
    uint32_t mo = 0, dy = 0, ms = 0;
    if (valid) {
      for (uint32_t j = 0; j < 4; j++) {
        if constexpr (!RH_TRUST) mo |= src.ld1(spos + j) << (8 * j);
        dy |= src.ld1(spos + 4 + j) << (8 * j);
        ms |= src.ld1(spos + 8 + j) << (8 * j);
      }
    }
    RH_REJECT(L, valid && mo != 0, E_DURATION, (int64_t)mo);
    valid = valid && L.live;
    lo = (uint64_t)dy * 86400000ull + (uint64_t)ms;
  } else if (op.a == BN_UUID_STR) {
    bool badtxt = false;
    if (valid) {
      uint32_t nn = 0;
      for (uint32_t j = 0; j < len; j++) {
        const uint32_t ch = src.ld1(spos + j);
        if (len == 36u && (j == 8 || j == 13 || j == 18 || j == 23)) { badtxt |= ch != '-'; continue; }
        const int h = hex_nibble(ch);
        badtxt |= h < 0;
        // big-endian byte order of the text; Arrow FixedSizeBinary keeps it: byte i of the value = hex pair i
        const uint32_t byte = nn >> 1, sh = (nn & 1) ? 0u : 4u;
        const uint64_t bits = (uint64_t)(h & 15) << (8 * (byte & 7) + sh);
        if (byte < 8) lo |= bits; else hi |= bits;
        nn++;
      }
      badtxt |= nn != 32u;
    }
    RH_REJECT(L, valid && badtxt, E_UUID);
    valid = valid && L.live;
  }
  L.cur += valid ? len : 0u;
  uint32_t row = 0;
  if (EMIT) {
    row = row_of(c, op.dom);
    if (act && W) {
      const uint64_t off = (uint64_t)row * W;
      if (op.a == BN_FIXED) {
        if (valid) copy_bytes<Ctx::kWide>(c.buf(op.buf1), off, src, spos, len);
        else fill_zero<Ctx::kWide>(c.buf(op.buf1), off, W);
      } else {
        if (!valid) { lo = 0; hi = 0; }
        st_at<u64u, Ctx::kWide>(c.buf(op.buf1), off, lo);
        if (W > 8) st_at<u64u, Ctx::kWide>(c.buf(op.buf1), off + 8, hi);       // (Duration(ms): 8 bytes per row)
      }
    }
  }
  put_validity<EMIT, Src::kSlide>(c, op, act, valid, row);
}

// NullableRecord (482-485 + 595-616): a null record null-fills its children
template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_rec_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  L.pstk = (L.pstk << 1) | (stk_t)(L.pres ? 1u : 0u);
  const bool dec = act && L.pres;
  int64_t dummy = 0;
  const bool isval = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, true, (op.flags & F_NULL_FIRST) != 0, false, false, dummy);
  const bool valid = isval && L.live;
  put_validity<EMIT, Src::kSlide>(c, op, act, valid, EMIT ? row_of(c, op.dom) : 0);
  L.pres = valid;
}
__device__ __forceinline__ void h_rec_end(Lane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
}

// UnionDecoder::decode / append_null (643-668): selected variant decodes, every other one null-fills
template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_union_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  L.pstk = (L.pstk << 1) | (stk_t)(L.pres ? 1u : 0u);
  L.sstk = (L.sstk << 8) | (sel_t)0xFFull;
  const bool dec = act && L.pres;
  int64_t idx = 0;
  void* const pu1 = EMIT ? c.buf(op.buf1) : nullptr;           // requested ahead of the head: see h_string
  const bool got = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, false, false, true, false, idx, 1) && L.live;
  const bool oor = got && (CAREFUL ? (idx < 0 || idx >= (int64_t)op.a) : (uint32_t)idx >= (uint32_t)op.a);
  RH_REJECT(L, oor, E_UNION, idx);
  uint32_t tidv = 0;
  if (got && L.live) { tidv = (uint32_t)idx; L.sstk = (L.sstk & ~(sel_t)0xFFull) | (sel_t)(uint64_t)idx; }
  if (EMIT && act) st_global<int8_t, Ctx::kWide>(pu1, row_of(c, op.dom), (int8_t)tidv);
}
__device__ __forceinline__ void h_variant(Lane& L, const Op& op) {
  L.pres = (L.pstk & 1) && ((uint32_t)(L.sstk & 0xFF) == (uint32_t)op.a);
}
__device__ __forceinline__ void h_union_end(Lane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
  L.sstk >>= 8;
}

// Between the columns of a wide schema (specialize.cpp): a source whose window moves keeps it under the cursors (SlideSrc::refill)
template <class Src, class Ctx>
__device__ __forceinline__ void h_topup(const Ctx& c, const Src& src, Lane& L) {
  if constexpr (Src::kSlide) src.refill(L, c.lane);
}

// ListDecoder / MapDecoder (+ Nullable*), 487-496, 703-770
template <bool EMIT, bool CAREFUL, int LA = 0, class Src, class Ctx>
__device__ __forceinline__ void h_list_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live;
  L.pstk = (L.pstk << 1) | (stk_t)(L.pres ? 1u : 0u);
  L.lstk = (L.lstk << 1) | (stk_t)(L.live ? 1u : 0u);
  const bool dec = act && L.pres;
  int64_t dummy = 0;
  const bool isval = read_head<CAREFUL, RH_TRUST, (CAREFUL ? 0 : LA)>(src, L, dec, (op.flags & F_NULLABLE) != 0, (op.flags & F_NULL_FIRST) != 0, false, false, dummy);
  const bool valid = isval && L.live;
  put_validity<EMIT, Src::kSlide>(c, op, act, valid, EMIT ? row_of(c, op.dom) : 0);
  L.live = valid;      // only rows that really carry a list enter the block loop
  L.pres = valid;
  c.remaining(op.c) = 0;
}

// read_block_count (689-700), exact form: negative counts carry a byte size, i64::MIN is an empty block
template <class Src, class Ctx>
__device__ __forceinline__ void list_next_slow(const Ctx& c, const Src& src, Lane& L, const Op& op, uint32_t& rm) {
  for (;;) {
    int64_t n = 0;
    uint32_t e = rd_varint(src, L.cur, L.end, n);
    if (e) { fail(L, e); return; }
    if (n < 0) {
      int64_t bsz;
      e = rd_varint(src, L.cur, L.end, bsz);   // block byte size, ignored
      if (e) { fail(L, e); return; }
      n = (int64_t)(0 - (uint64_t)n);
    }
    if (n == 0) { L.live = false; return; }
    if (n < 0) continue;                         // i64::MIN negates to itself: `0..n` is empty
    // Clamp the trip count: with m = min wire bytes per item and R bytes left, no more than R/m
    // items can decode, so item R/m+1 raises the same error the reference hits.
    const uint64_t R = L.end - L.cur;
    if (op.buf2 /*min wire bytes per item*/ > 0) {
      const uint64_t cap = R / (uint32_t)op.buf2 + 1;
      rm = (uint32_t)((uint64_t)n < cap ? (uint64_t)n : cap);
    } else if ((uint64_t)n > 0x00FFFFFFull) {
      fail(L, E_LIST_RANGE, n);
    } else {
      rm = (uint32_t)n;
    }
    return;
  }
}

// Head of the block loop.  Returns true while any lane of the wave still has an item (wave-uniform).
template <bool CAREFUL, bool TRUST = false, class Src, class Ctx>
__device__ __forceinline__ bool h_list_next(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  if constexpr (Src::kSlide) src.refill(L, c.lane);          // a record larger than the window: move the window up to the cursor
  uint32_t& rm = c.remaining(op.c);
  if constexpr (!CAREFUL && !TRUST) L.live = L.live && !L.redo;      // a lane that has met an anomaly (RH_REJECT_SOFT) stops iterating here
  const bool need = L.live && rm == 0;            // this lane is at a block boundary
  if (TRUST) {               // every block header of this tile took the one-read form below in the size pass, unclamped
    uint32_t raw, n;
    (void)varint32(src.ld4(L.cur), 4u, raw, n);      // (a block count: the 4-byte form -- an array of two million items is met in production)
    if (need) {
      L.cur += n;
      if ((raw >> 1) == 0) L.live = false;
      else rm = raw >> 1;
    }
    if (!__any(L.live)) return false;
    L.pres = L.live;
    return true;
  }
  // common wire form: a small positive count, or the 0 terminator, in one byte..four bytes
  const uint32_t x = src.ld4(L.cur);
  uint32_t raw, n;
  // bytes left, 0 for a cursor that is already past its record's end (only the fast walk can be: read_head): such a
  // lane fails here, so a garbage block count never starts a loop
  const int32_t left = (int32_t)(L.end - L.cur);
  const uint32_t lav = CAREFUL ? (uint32_t)left : (uint32_t)(left < 0 ? 0 : left);
  const bool okv = varint32(x, lav, raw, n);
  const bool fast = need && okv && (raw & 1u) == 0 && (op.buf2 > 0 || raw == 0);   // non-negative; zero-width items take the exact path
  const bool slow = need && !fast;
  if (fast) {
    const uint32_t cnt = raw >> 1;
    L.cur += n;
    if (cnt == 0) L.live = false;
    else {
      const uint32_t cap = (lav - n) / (uint32_t)(op.buf2 > 0 ? op.buf2 : 1) + 1;
      rm = cnt < cap ? cnt : cap;
    }
  }
  if (CAREFUL) {
    if (__any(slow)) {
      if (slow) list_next_slow(c, src, L, op, rm);
    }
  } else {
    reject<false>(L, slow, 0);
  }
  if (!__any(L.live)) return false;
  L.pres = L.live;
  return true;
}

template <class Ctx>
__device__ __forceinline__ void h_list_tail(const Ctx& c, Lane& L, const Op& op) {
  if (L.live) {
    c.remaining(op.c) -= 1;
    c.counter(op.a - 1) += 1;     // op.a = child row domain
  }
}

template <bool EMIT, class Ctx>
__device__ __forceinline__ void h_list_end(const Ctx& c, Lane& L, const Op& op) {
  L.live = L.lstk & 1;
  L.lstk >>= 1;
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
  if (EMIT && L.live) {
    // cumulative child rows so far == Arrow offset of the next row (null / empty rows repeat it)
    st_global<uint32_t, Ctx::kWide>(c.buf(op.buf1), row_of(c, op.dom) + 1, c.gbase(op.a - 1) + c.counter(op.a - 1));
  }
}

}  // namespace rh
