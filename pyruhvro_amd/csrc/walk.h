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

// --------------------------------------------------------------------------
// wave primitives (wave = 64 lanes)
// --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// --------------------------------------------------------------------------
// byte sources.  Positions are relative to the workgroup's window base.
// gfx950 services unaligned DS / global accesses in hardware, so an 8-byte
// read at any byte position is ONE ds_read_b64 / global_load_dwordx2.
// --------------------------------------------------------------------------
struct LdsSrc {
  const uint8_t* w;   // LDS window; >= 8 readable bytes past the last record
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const { return w[p]; }
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const { return *reinterpret_cast<const u64u*>(w + p); }
};

struct GlobalSrc {
  const uint8_t* g;   // payload + window base
  uint64_t lim;       // readable bytes from g
  __device__ __forceinline__ uint32_t ld1(uint32_t p) const { return g[p]; }
  __device__ __forceinline__ uint64_t ld8(uint32_t p) const {
    if ((uint64_t)p + 8 <= lim) return *reinterpret_cast<const u64u*>(g + p);
    uint64_t x = 0;
    for (uint32_t j = 0; j < 8 && (uint64_t)p + j < lim; j++) x |= (uint64_t)g[p + j] << (8 * j);
    return x;
  }
};

// Arrow buffers live in HBM: typed global-address-space accessors keep the compiler from emitting
// flat_* instructions (which tie up both the vector-memory and the LDS counters).
#define RH_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ void st_global(void* base, uint64_t idx, T v) {
  reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(base))[idx] = v;
}
__device__ __forceinline__ void atomic_or_global(void* base, uint64_t idx, uint32_t bits) {
  __hip_atomic_fetch_or(reinterpret_cast<RH_GLOBAL uint32_t*>(reinterpret_cast<uintptr_t>(base)) + idx, bits,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// copy `len` bytes window[sp..] -> d, any alignment on both sides: 8 bytes per DS / global instruction.
#define RH_COPY_BODY(P64, P32, P16, P8)                                         \
  uint32_t j = 0;                                                                \
  for (; j + 8 <= len; j += 8) *reinterpret_cast<P64>(d + j) = s.ld8(sp + j);    \
  if (len & 7) {                                                                 \
    uint64_t x = s.ld8(sp + j);                                                  \
    if (len & 4) { *reinterpret_cast<P32>(d + j) = (uint32_t)x; x >>= 32; j += 4; } \
    if (len & 2) { *reinterpret_cast<P16>(d + j) = (uint16_t)x; x >>= 16; j += 2; } \
    if (len & 1) *reinterpret_cast<P8>(d + j) = (uint8_t)x;                      \
  }

template <class Src>   // d in LDS (string staging area)
__device__ __forceinline__ void copy_bytes(uint8_t* d, const Src& s, uint32_t sp, uint32_t len) {
  RH_COPY_BODY(u64u*, u32u*, u16u*, uint8_t*)
}
template <class Src>   // d in HBM (unstaged fallback)
__device__ __forceinline__ void copy_bytes(RH_GLOBAL uint8_t* d, const Src& s, uint32_t sp, uint32_t len) {
  RH_COPY_BODY(RH_GLOBAL u64u*, RH_GLOBAL u32u*, RH_GLOBAL u16u*, RH_GLOBAL uint8_t*)
}
#undef RH_COPY_BODY

template <class D>
__device__ __forceinline__ void copy_plain(D* d, const uint8_t* s, uint32_t len) {
  for (uint32_t j = 0; j < len; j++) d[j] = s[j];
}

// --------------------------------------------------------------------------
// per-lane walker state
// --------------------------------------------------------------------------
struct Lane {
  uint32_t cur, end;   // byte cursor / record end, relative to the window base (cur <= end always)
  uint32_t err;        // ErrCode, 0 = ok
  int64_t edetail;
  bool live, pres;
  uint32_t pstk;       // saved `pres` bits   (nullable record / union / list)
  uint32_t lstk;       // saved `live` bits   (list)
  uint64_t sstk;       // saved union selectors, 8 bits each
};

// --------------------------------------------------------------------------
// primitive readers (fast_decode.rs:845-922)
// --------------------------------------------------------------------------
// byte-at-a-time form with the reference's exact error order (854-869)
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

// Decode a zig-zag LEB128 varint that starts at bit 0 of x (x holds `nx` valid bytes, `avail` bytes
// remain in the record).  Branch-free for varints of <= nx bytes; returns false when the slow path
// must decide (longer varint, or one that runs past the record end).
__device__ __forceinline__ bool varint_from_bits(uint64_t x, uint32_t nx, uint32_t avail, int64_t& out, uint32_t& n) {
  const uint64_t t = ~x & 0x8080808080808080ull;   // bit 7 of every byte WITHOUT a continuation flag
  if (t == 0) return false;
  n = ((uint32_t)__builtin_ctzll(t) >> 3) + 1;
  if (n > nx || n > avail) return false;
  x &= t ^ (t - 1);                                // keep bytes 0..n-1
  x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
  x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
  x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
  out = (int64_t)(x >> 1) ^ -(int64_t)(x & 1);
  return true;
}

template <class Src>
__device__ __forceinline__ uint32_t rd_varint(const Src& src, uint32_t& cur, uint32_t end, int64_t& out) {
  uint32_t n;
  if (varint_from_bits(src.ld8(cur), 8, end - cur, out, n)) { cur += n; return E_OK; }
  return rd_varint_slow(src, cur, end, out);
}

// union_branch (585-593) followed by the value's leading varint, from ONE 8-byte read when both fit.
// Returns isval; when isval and `want_varint`, v holds the varint that follows the branch.
template <class Src>
__device__ __forceinline__ bool rd_branch_then_varint(const Src& src, Lane& L, bool null_first, bool want_varint, int64_t& v) {
  const uint64_t x = src.ld8(L.cur);
  const uint32_t avail = L.end - L.cur;
  const uint32_t b0 = (uint32_t)x & 0xFF;
  if (avail >= 1 && (b0 == 0 || b0 == 2)) {        // branch 0 / 1, single byte
    const bool isval = (b0 == 0) ? !null_first : null_first;
    L.cur += 1;
    if (isval && want_varint) {
      uint32_t n;
      if (varint_from_bits(x >> 8, 7, avail - 1, v, n)) L.cur += n;
      else {
        uint32_t e = rd_varint_slow(src, L.cur, L.end, v);
        if (e) L.err = e;
      }
    }
    return isval;
  }
  int64_t idx = 0;
  uint32_t e = rd_varint_slow(src, L.cur, L.end, idx);
  if (e) { L.err = e; return false; }
  if (idx != 0 && idx != 1) { L.err = E_BRANCH; L.edetail = idx; return false; }
  const bool isval = (idx == 0) ? !null_first : null_first;
  if (isval && want_varint) {
    e = rd_varint(src, L.cur, L.end, v);
    if (e) L.err = e;
  }
  return isval;
}

// --------------------------------------------------------------------------
// shared output helpers.  Ctx provides:
//   uint32_t& counter(int id)   per-lane counter (child rows / string bytes), block-local after the scan
//   uint32_t& remaining(int d)  items left in the current block of list depth d
//   void* buf(int id)           this chunk's Arrow buffer `id`
//   uint32_t gbase(int id)      chunk-relative base of this workgroup for counter id
//   uint32_t stage_off(int id)  LDS staging offset of string counter id (kNoStage = write to HBM directly)
//   uint8_t* stage()            LDS staging area
//   void add_nulls(int node, uint32_t n)
//   lrow, lane, wave_live, sym_off, sym_data
// --------------------------------------------------------------------------
constexpr uint32_t kNoStage = 0xFFFFFFFFu;

template <class Ctx>
__device__ __forceinline__ uint32_t row_of(const Ctx& c, int dom) {
  return dom == 0 ? c.lrow : c.gbase(dom - 1) + c.counter(dom - 1);
}

// validity bit + null count of one row (the buffer exists iff F_CAN_NULL)
template <bool EMIT, class Ctx>
__device__ __forceinline__ void put_validity(const Ctx& c, const Op& op, bool act, bool valid, uint32_t row) {
  if (!EMIT) return;
  if (!(op.flags & F_CAN_NULL)) return;
  if (op.dom == 0) {
    const uint64_t m = __ballot(valid);
    const uint64_t nm = __ballot(act && !valid);
    if (c.lane == 0 && c.wave_live) {
      st_global<uint64_t>(c.buf(op.buf0), c.lrow >> 6, m);
      if (nm) c.add_nulls(op.node, (uint32_t)__popcll(nm));
    }
  } else if (act) {
    if (valid) atomic_or_global(c.buf(op.buf0), row >> 5, 1u << (row & 31));
    else c.add_nulls(op.node, 1u);
  }
}

// --------------------------------------------------------------------------
// field handlers
// --------------------------------------------------------------------------
// int/long/float/double/boolean/date/timestamp leaf, optionally Nullable* (424-432, 434-473)
template <bool EMIT, class Src, class Ctx>
__device__ __forceinline__ void h_fixed(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  const bool dec = act && L.pres;
  const bool is_int = op.a == FK_I32 || op.a == FK_I64;
  bool isval = dec;
  int64_t v = 0;
  bool have_v = false;
  if ((op.flags & F_NULLABLE) && dec) {
    isval = rd_branch_then_varint(src, L, (op.flags & F_NULL_FIRST) != 0, is_int, v);
    have_v = true;
  }
  uint64_t bits = 0;
  if (dec && isval && L.err == 0) {
    if (is_int) {
      if (!have_v) {
        uint32_t e = rd_varint(src, L.cur, L.end, v);
        if (e) L.err = e;
      }
      bits = op.a == FK_I32 ? (uint64_t)(uint32_t)(int32_t)v : (uint64_t)v;   // `as i32` truncates (424,430)
    } else if (op.a == FK_F32) {
      if (L.end - L.cur < 4) L.err = E_EOB_F32;
      else { bits = (uint32_t)src.ld8(L.cur); L.cur += 4; }
    } else if (op.a == FK_F64) {
      if (L.end - L.cur < 8) L.err = E_EOB_F64;
      else { bits = src.ld8(L.cur); L.cur += 8; }
    } else {   // FK_BOOL, 893-900
      if (L.cur >= L.end) L.err = E_EOB;
      else {
        uint32_t b = src.ld1(L.cur++);
        if (b > 1) { L.err = E_BOOL; L.edetail = b; }
        bits = b;
      }
    }
  }
  const bool valid = dec && isval && L.err == 0;
  if (!valid) bits = 0;   // zero under nulls (arrow-rs append_null)
  uint32_t row = 0;
  if (EMIT) {
    row = row_of(c, op.dom);
    if (op.a == FK_BOOL) {
      if (op.dom == 0) {
        const uint64_t m = __ballot(bits != 0);
        if (c.lane == 0 && c.wave_live) st_global<uint64_t>(c.buf(op.buf1), c.lrow >> 6, m);
      } else if (act && bits) {
        atomic_or_global(c.buf(op.buf1), row >> 5, 1u << (row & 31));
      }
    } else if (act) {
      if (op.a == FK_I32 || op.a == FK_F32) st_global<uint32_t>(c.buf(op.buf1), row, (uint32_t)bits);
      else st_global<uint64_t>(c.buf(op.buf1), row, bits);
    }
  }
  put_validity<EMIT>(c, op, act, valid, row);
}

// string leaf / map key (429, 454-457, 752, read_string 902-922) and enum -> symbol text (570-578)
template <bool EMIT, class Src, class Ctx>
__device__ __forceinline__ void h_string(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  const bool dec = act && L.pres;
  bool isval = dec;
  int64_t v = 0;
  bool have_v = false;
  if ((op.flags & F_NULLABLE) && dec) {
    isval = rd_branch_then_varint(src, L, (op.flags & F_NULL_FIRST) != 0, true, v);
    have_v = true;
  }
  uint32_t len = 0, spos = 0;
  if (dec && isval && L.err == 0) {
    if (!have_v) {
      uint32_t e = rd_varint(src, L.cur, L.end, v);
      if (e) L.err = e;
    }
    if (L.err == 0) {
      if (op.code == OP_STRING) {
        if (v < 0) L.err = E_NEGLEN;
        else if ((uint64_t)(L.end - L.cur) < (uint64_t)v) L.err = E_EOB_STR;
        else { len = (uint32_t)v; spos = L.cur; L.cur += len; }
      } else {
        if ((uint64_t)v >= (uint64_t)op.c) { L.err = E_ENUM; L.edetail = v; }
        else {
          spos = c.sym_off[op.b + (int32_t)v];
          len = c.sym_off[op.b + (int32_t)v + 1] - spos;
        }
      }
    }
  }
  const bool valid = dec && isval && L.err == 0;
  if (!valid) len = 0;
  uint32_t& bo = c.counter(op.a);
  const uint32_t o = bo;
  uint32_t row = 0;
  if (EMIT) {
    row = row_of(c, op.dom);
    if (act) {
      const uint32_t gb = c.gbase(op.a);
      st_global<uint32_t>(c.buf(op.buf1), (uint64_t)row + 1, gb + o + len);   // offsets repeat under nulls
      if (len) {
        const uint32_t so = c.stage_off(op.a);
        if (so != kNoStage) {
          uint8_t* d = c.stage() + so + o;
          if (op.code == OP_STRING) copy_bytes(d, src, spos, len);
          else copy_plain(d, c.sym_data + spos, len);
        } else {
          RH_GLOBAL uint8_t* d = reinterpret_cast<RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(c.buf(op.buf2))) + gb + o;
          if (op.code == OP_STRING) copy_bytes(d, src, spos, len);
          else copy_plain(d, c.sym_data + spos, len);
        }
      }
    }
  }
  if (act) bo = o + len;
  put_validity<EMIT>(c, op, act, valid, row);
}

// NullableRecord (482-485 + 595-616): a null record null-fills its children
template <bool EMIT, class Src, class Ctx>
__device__ __forceinline__ void h_rec_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  const bool dec = act && L.pres;
  bool isval = dec;
  int64_t dummy;
  if (dec) isval = rd_branch_then_varint(src, L, (op.flags & F_NULL_FIRST) != 0, false, dummy);
  const bool valid = dec && isval && L.err == 0;
  put_validity<EMIT>(c, op, act, valid, EMIT ? row_of(c, op.dom) : 0);
  L.pres = valid;
}
__device__ __forceinline__ void h_rec_end(Lane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
}

// UnionDecoder::decode / append_null (643-668): selected variant decodes, every other one null-fills
template <bool EMIT, class Src, class Ctx>
__device__ __forceinline__ void h_union_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  L.sstk = (L.sstk << 8) | 0xFFull;
  const bool dec = act && L.pres;
  uint32_t tidv = 0;
  if (dec) {
    int64_t idx = 0;
    uint32_t e = rd_varint(src, L.cur, L.end, idx);
    if (e) L.err = e;
    else if (idx < 0 || idx >= (int64_t)op.a) { L.err = E_UNION; L.edetail = idx; }
    else { tidv = (uint32_t)idx; L.sstk = (L.sstk & ~0xFFull) | (uint64_t)idx; }
  }
  if (EMIT && act) st_global<int8_t>(c.buf(op.buf1), row_of(c, op.dom), (int8_t)tidv);
}
__device__ __forceinline__ void h_variant(Lane& L, const Op& op) {
  L.pres = (L.pstk & 1) && ((uint32_t)(L.sstk & 0xFF) == (uint32_t)op.a);
}
__device__ __forceinline__ void h_union_end(Lane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
  L.sstk >>= 8;
}

// ListDecoder / MapDecoder (+ Nullable*), 487-496, 703-770
template <bool EMIT, class Src, class Ctx>
__device__ __forceinline__ void h_list_begin(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  L.lstk = (L.lstk << 1) | (L.live ? 1u : 0u);
  const bool dec = act && L.pres;
  bool isval = dec;
  int64_t dummy;
  if ((op.flags & F_NULLABLE) && dec) isval = rd_branch_then_varint(src, L, (op.flags & F_NULL_FIRST) != 0, false, dummy);
  const bool valid = dec && isval && L.err == 0;
  put_validity<EMIT>(c, op, act, valid, EMIT ? row_of(c, op.dom) : 0);
  L.live = valid;      // only rows that really carry a list enter the block loop
  L.pres = valid;
  c.remaining(op.c) = 0;
}

// read_block_count (689-700).  Returns true while any lane of the wave still has an item (wave-uniform).
template <class Src, class Ctx>
__device__ __forceinline__ bool h_list_next(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const bool act = L.live && L.err == 0;
  uint32_t& rm = c.remaining(op.c);
  if (act && rm == 0) {
    for (;;) {
      int64_t n = 0;
      uint32_t e = rd_varint(src, L.cur, L.end, n);
      if (e) { L.err = e; break; }
      if (n < 0) {
        int64_t bsz;
        e = rd_varint(src, L.cur, L.end, bsz);   // block byte size, ignored
        if (e) { L.err = e; break; }
        n = (int64_t)(0 - (uint64_t)n);
      }
      if (n == 0) { L.live = false; break; }
      if (n < 0) continue;                         // i64::MIN negates to itself: `0..n` is empty
      // Clamp the trip count: with m = min wire bytes per item and R bytes left, no more than R/m
      // items can decode, so item R/m+1 raises the same error the reference hits.
      const uint64_t R = L.end - L.cur;
      if (op.buf2 /*min wire bytes per item*/ > 0) {
        const uint64_t cap = R / (uint32_t)op.buf2 + 1;
        rm = (uint32_t)((uint64_t)n < cap ? (uint64_t)n : cap);
      } else if ((uint64_t)n > 0x00FFFFFFull) {
        L.err = E_LIST_RANGE; L.edetail = n;
      } else {
        rm = (uint32_t)n;
      }
      break;
    }
  }
  const bool item = L.live && L.err == 0;
  if (!__any(item)) return false;
  L.pres = L.live;
  return true;
}

template <class Ctx>
__device__ __forceinline__ void h_list_tail(const Ctx& c, Lane& L, const Op& op) {
  if (L.live && L.err == 0) {
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
  if (EMIT && L.live && L.err == 0) {
    // cumulative child rows so far == Arrow offset of the next row (null / empty rows repeat it)
    st_global<uint32_t>(c.buf(op.buf1), (uint64_t)row_of(c, op.dom) + 1, c.gbase(op.a - 1) + c.counter(op.a - 1));
  }
}

}  // namespace rh
