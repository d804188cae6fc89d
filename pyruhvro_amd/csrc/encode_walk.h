// Arrow -> Avro encode on CDNA4 (gfx950): the per-row walk, shared by the generic schema-program interpreter
// (encode.hip) and the schema-specialised kernels that specialize.cpp generates and hiprtc compiles.
//
// Reference: ruhvro/src/fast_encode.rs:387-599 (per-row write of every encoder variant, zig-zag varints, one block
// per array/map) and ruhvro/src/serialize.rs:19-67 (chunking).  One lane = one row of the batch, one workgroup =
// 256 consecutive rows of one output chunk.  The schema program is the decoder's (program.h); its buffer ids name
// the INPUT Arrow buffers here, every one rebased to logical row 0 by the host (engine.cpp, EncodeBinder).
//
// Every node is handled in two halves so that the specialised kernels can issue the loads of many fields before
// the first byte is produced (the walk is bound by dependent-load latency, not by bytes):
//   e_*_load   what the node needs from its input buffers at the lane's current row -- UNCONDITIONAL loads: every
//              lane, present or not, reads a valid address (the host pads every buffer, points absent validity
//              bitmaps at an all-ones bitmap, and row cursors never leave [0, rows]);
//   e_*_put    fast_encode.rs's write for that node, predicated on the lane being present.
#pragma once
#include "encode.h"
#include "kernel_common.h"

namespace rh {

#define RH_LDS __attribute__((address_space(3)))   // typed LDS pointers: ds_* instructions, never flat_*

constexpr int M_SIZE = 0, M_DIRECT = 1, M_STAGED = 2;   // what a walk does with the bytes it produces

// Phase timing (RUHVRO_HIP_PROFILE=1 builds): lane 0 of every wave adds the shader-clock cycles between marks to
// P.prof[slot]; the host prints per-wave means.  Compiled out otherwise.
#ifdef RH_PROFILE
struct PhaseClock {
  unsigned long long pd[24], tprev;
  __device__ __forceinline__ void init() {
    for (int i = 0; i < 24; i++) pd[i] = 0;
    tprev = clock64();
  }
  __device__ __forceinline__ void mark(int slot) {
    const unsigned long long t = clock64();
    pd[slot] += t - tprev;
    tprev = t;
  }
  __device__ __forceinline__ void flush(unsigned long long* prof) const {
    if ((threadIdx.x & 63) == 0 && prof) {
      unsigned long long* row = prof + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63) * 32;
      for (int i = 0; i < 24; i++)
        if (pd[i]) atomicAdd(&row[i], pd[i]);
    }
  }
};
#else
struct PhaseClock {
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush(unsigned long long*) const {}
};
#endif

struct ELane {
  PhaseClock clk;
  uint32_t len;        // bytes produced so far by this row (size pass: counted; emit pass: cursor)
  uint32_t err;
  int64_t edetail;
  uint32_t eop;
  bool live, pres;
  uint32_t pstk, lstk;
  uint64_t sstk;
  __device__ __forceinline__ bool writes() const { return live && pres && err == 0; }
};

struct FixedV { uint64_t bits; bool valid; };      // raw value (ints sign-extended to 64 bits), validity bit
struct SpanV { uint32_t s0, s1; bool valid; };     // offsets[row], offsets[row+1], validity bit


// ---- byte sinks ------------------------------------------------------------------------------------------------
template <int MODE, class Ctx>
__device__ __forceinline__ void put_byte(const Ctx& c, ELane& L, uint8_t b) {
  if (MODE == M_DIRECT) c.out[L.len] = b;
  if (MODE == M_STAGED) c.lout[L.len] = b;
  L.len++;
}

// write_zigzag_long, fast_encode.rs:583-591
template <int MODE, class Ctx>
__device__ __forceinline__ void put_varint(const Ctx& c, ELane& L, int64_t v) {
  uint64_t zz = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  if (MODE == M_SIZE) {                       // bytes = ceil(significant bits / 7), at least 1
    const uint32_t bits = 64u - (uint32_t)__builtin_clzll(zz | 1ull);
    L.len += (bits + 6u) / 7u;
    return;
  }
  for (;;) {
    const bool more = (zz & ~0x7Full) != 0;
    put_byte<MODE>(c, L, (uint8_t)((zz & 0x7F) | (more ? 0x80 : 0)));
    if (!more) break;
    zz >>= 7;
  }
}

// N little-endian bytes of a register (float / double payloads, fast_encode.rs:419-431)
template <int MODE, int N, class Ctx>
__device__ __forceinline__ void put_raw(const Ctx& c, ELane& L, uint64_t bits) {
  if (MODE == M_DIRECT) {
    if (N == 4) *reinterpret_cast<RH_GLOBAL u32u*>(c.out + L.len) = (uint32_t)bits;
    else *reinterpret_cast<RH_GLOBAL u64u*>(c.out + L.len) = bits;
  }
  if (MODE == M_STAGED) {
    RH_LDS uint8_t* d = c.lout + L.len;
    if (((uint32_t)(uintptr_t)d & 3u) == 0) {          // aligned dwords when the cursor allows it, bytes otherwise
      volatile RH_LDS uint32_t* dw = reinterpret_cast<volatile RH_LDS uint32_t*>(d);   // volatile: two dword stores, not one b64
      dw[0] = (uint32_t)bits;
      if (N == 8) dw[1] = (uint32_t)(bits >> 32);
    } else {
#pragma unroll
      for (int j = 0; j < N; j++) d[j] = (uint8_t)(bits >> (8 * j));
    }
  }
  L.len += N;
}

// m <= 32 bytes held in w[0..8) (w[8] = 0) -> LDS at D, any alignment, exactly m bytes: byte head up to the next
// dword boundary, aligned dwords funnel-shifted out of the register stream (v_alignbyte), byte tail.  LDS wants
// aligned accesses -- unaligned DS accesses are serviced one lane per cycle on this part.
__device__ __forceinline__ void lds_put_32(RH_LDS uint8_t* D, const uint32_t (&w)[9], uint32_t m) {
  const uint32_t h0 = (4u - ((uint32_t)(uintptr_t)D & 3u)) & 3u;
  const uint32_t h = h0 < m ? h0 : m;
  if (h > 0) D[0] = (uint8_t)w[0];
  if (h > 1) D[1] = (uint8_t)(w[0] >> 8);
  if (h > 2) D[2] = (uint8_t)(w[0] >> 16);
  const uint32_t q = (m - h) >> 2, t = (m - h) & 3u;
  volatile RH_LDS uint32_t* Dw = reinterpret_cast<volatile RH_LDS uint32_t*>(D + h);   // volatile: keep them dword stores
  uint32_t xt = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t x = __builtin_amdgcn_alignbyte(w[k + 1], w[k], h);        // stream bytes [4k+h, 4k+h+4)
    if ((uint32_t)k < q) Dw[k] = x;
    if ((uint32_t)k == q) xt = x;
  }
  RH_LDS uint8_t* Dt = D + h + 4u * q;
  if (t > 0) Dt[0] = (uint8_t)xt;
  if (t > 1) Dt[1] = (uint8_t)(xt >> 8);
  if (t > 2) Dt[2] = (uint8_t)(xt >> 16);
}

// first 32 bytes at s (any alignment; the host pads every data buffer so the over-read stays inside the arena)
__device__ __forceinline__ void ld_32(const RH_GLOBAL uint8_t* s, uint32_t (&w)[9]) {
  const v4w a = *reinterpret_cast<const RH_GLOBAL v4wu*>(s);
  const v4w b = *reinterpret_cast<const RH_GLOBAL v4wu*>(s + 16);
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; w[8] = 0;
}

// The first 32 bytes of a string, fetched ahead of its put (e_string_fetch): a string costs the walk no HBM round trip
// of its own when its bytes were requested together with those of the row's other strings.
struct Str32 { uint32_t w[9]; };

// put_bytes with bytes [0, 32) already in registers
template <int MODE, class Ctx>
__device__ __forceinline__ void put_bytes_pf(const Ctx& c, ELane& L, const RH_GLOBAL uint8_t* s, uint32_t n, const Str32& pf) {
  if (MODE == M_DIRECT) {
    RH_GLOBAL uint8_t* d = c.out + L.len;
    uint32_t j = 0;
    for (; j + 8 <= n; j += 8) *reinterpret_cast<RH_GLOBAL u64u*>(d + j) = *reinterpret_cast<const RH_GLOBAL u64u*>(s + j);
    for (; j < n; j++) d[j] = s[j];
  }
  if (MODE == M_STAGED) {
    RH_LDS uint8_t* d = c.lout + L.len;
    if (n > 0) lds_put_32(d, pf.w, n < 32u ? n : 32u);
    for (uint32_t j = 32; j < n; j += 32) {
      uint32_t w[9];
      ld_32(s + j, w);
      const uint32_t m = n - j;
      lds_put_32(d + j, w, m < 32u ? m : 32u);
    }
  }
  L.len += n;
}

template <int MODE, class Ctx>
__device__ __forceinline__ void put_bytes(const Ctx& c, ELane& L, const RH_GLOBAL uint8_t* s, uint32_t n) {
  if (MODE == M_DIRECT) {
    RH_GLOBAL uint8_t* d = c.out + L.len;
    uint32_t j = 0;
    for (; j + 8 <= n; j += 8) *reinterpret_cast<RH_GLOBAL u64u*>(d + j) = *reinterpret_cast<const RH_GLOBAL u64u*>(s + j);
    for (; j < n; j++) d[j] = s[j];
  }
  if (MODE == M_STAGED) {       // 32 bytes per round trip to HBM: two 16-byte loads in flight, then register -> LDS
    RH_LDS uint8_t* d = c.lout + L.len;
    for (uint32_t j = 0; j < n; j += 32) {
      uint32_t w[9];
      ld_32(s + j, w);
      const uint32_t m = n - j;
      lds_put_32(d + j, w, m < 32u ? m : 32u);
    }
  }
  L.len += n;
}

// write_nullable (563-572): branch index for a null / a value -- zig-zag of 0 or 1 is one byte
template <int MODE, class Ctx>
__device__ __forceinline__ void put_branch(const Ctx& c, ELane& L, bool is_null, bool null_first) {
  put_byte<MODE>(c, L, (uint8_t)((is_null != null_first) ? 2 : 0));
}

// ---- loads -----------------------------------------------------------------------------------------------------
template <class Ctx>
__device__ __forceinline__ bool ld_bit(const Ctx& c, int buf, uint32_t r) {      // validity / boolean value of logical row r
  const uint32_t b = r + c.bitoff(buf);
  return (reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(buf))[b >> 3] >> (b & 7)) & 1;
}

template <class Ctx>
__device__ __forceinline__ FixedV e_fixed_load(const Ctx& c, const Op op, uint32_t ahead = 0) {      // ahead: rows in front of the cursor
  const uint32_t r = c.row(op.dom) + ahead;
  FixedV v;
  v.valid = (op.flags & F_NULLABLE) ? ld_bit(c, op.buf0, r) : true;
  if (op.a == FK_I32) v.bits = (uint64_t)(int64_t)reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1))[r];
  else if (op.a == FK_F32) v.bits = reinterpret_cast<const RH_GLOBAL uint32_t*>(c.in(op.buf1))[r];
  else if (op.a == FK_BOOL) v.bits = ld_bit(c, op.buf1, r) ? 1 : 0;
  else v.bits = reinterpret_cast<const RH_GLOBAL uint64_t*>(c.in(op.buf1))[r];
  return v;
}

template <class Ctx>
__device__ __forceinline__ SpanV e_span_load(const Ctx& c, const Op op, uint32_t ahead = 0) {        // string / enum / list / map offsets
  const uint32_t r = c.row(op.dom) + ahead;
  SpanV v;
  v.valid = (op.flags & F_NULLABLE) ? ld_bit(c, op.buf0, r) : true;
  const RH_GLOBAL uint32_t* off = reinterpret_cast<const RH_GLOBAL uint32_t*>(c.in(op.buf1));
  if (op.dom != 0) {      // rows of a child domain are scattered over the lanes: one 8-byte request per lane, not two
    const uint64_t o = *reinterpret_cast<const RH_GLOBAL u64u*>(off + r);     // dwords (for consecutive rows two dword loads win)
    v.s0 = (uint32_t)o;
    v.s1 = (uint32_t)(o >> 32);
    return v;
  }
  v.s0 = off[r];
  v.s1 = off[r + 1];
  return v;
}

template <class Ctx>
__device__ __forceinline__ bool e_rec_load(const Ctx& c, const Op op, uint32_t ahead = 0) { return ld_bit(c, op.buf0, c.row(op.dom) + ahead); }

template <class Ctx>
__device__ __forceinline__ int32_t e_union_load(const Ctx& c, const Op op, uint32_t ahead = 0) {
  return reinterpret_cast<const RH_GLOBAL int8_t*>(c.in(op.buf1))[c.row(op.dom) + ahead];
}

// SURVEY 8(f) N4 (beyond the reference, whose encoder gate is false for these types): fixed(N),
// decimal, uuid and duration leaves.  The 16 value bytes of a decimal / uuid row travel in registers; fixed(N) is copied from HBM.
struct BinV { uint64_t lo, hi; bool valid; };

template <class Ctx>
__device__ __forceinline__ BinV e_bin_load(const Ctx& c, const Op op, uint32_t ahead = 0) {
  const uint32_t r = c.row(op.dom) + ahead;
  BinV v;
  v.lo = 0; v.hi = 0;
  v.valid = (op.flags & F_NULLABLE) ? ld_bit(c, op.buf0, r) : true;
  if (op.a == BN_DURATION) {                                    // Duration(ms): one i64 per row
    v.lo = reinterpret_cast<const RH_GLOBAL u64u*>(c.in(op.buf1))[r];
  } else if (op.a != BN_FIXED) {                                // Decimal128 / FixedSizeBinary(16): 16 bytes per row
    const RH_GLOBAL u64u* p = reinterpret_cast<const RH_GLOBAL u64u*>(c.in(op.buf1)) + 2ull * r;
    v.lo = p[0]; v.hi = p[1];
  }
  return v;
}

__device__ __forceinline__ uint32_t bin_byte(const BinV& v, uint32_t j) {      // byte j of the 16 little-endian value bytes
  return (uint32_t)((j < 8 ? v.lo >> (8 * j) : v.hi >> (8 * (j - 8))) & 0xFFu);
}
__device__ __forceinline__ uint8_t hex_digit(uint32_t x) { return (uint8_t)(x < 10 ? '0' + x : 'a' + (x - 10)); }

// the Avro 1.11 wire forms (mirror of walk.h h_bin): fixed = the N bytes; decimal on bytes = length + minimal big-endian
// two's complement; decimal on fixed(N) = the low N bytes, big-endian; uuid on string = 36 characters of lower-case
// 8-4-4-4-12 hex text.  Not a hot path: bytes are produced one at a time.
template <int MODE, class Ctx>
__device__ __forceinline__ void e_bin_put(const Ctx& c, ELane& L, const Op op, const BinV v) {
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  if (op.a == BN_FIXED) {
    const uint32_t W = (uint32_t)op.c;
    put_bytes<MODE>(c, L, reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf1)) + (uint64_t)c.row(op.dom) * W, W);
  } else if (op.a == BN_DURATION) {
    // months = 0, days = v / 86 400 000 (at most 2^32-1), milliseconds = the rest (three little-endian u32): the split
    // h_bin's sum inverts, for every value it can produce.  A negative count, or one beyond 2^32-1 days + 2^32-1 ms, has
    // no wire form.
    uint64_t days = v.lo / 86400000ull;
    days = days > 0xFFFFFFFFull ? 0xFFFFFFFFull : days;
    const uint64_t ms = v.lo - days * 86400000ull;
    if ((int64_t)v.lo < 0 || ms > 0xFFFFFFFFull) { L.err = EE_DURATION; L.eop = 0; L.edetail = c.row(op.dom); return; }
    if (MODE == M_SIZE) { L.len += 12; return; }
    put_raw<MODE, 4>(c, L, 0ull);
    put_raw<MODE, 4>(c, L, days);
    put_raw<MODE, 4>(c, L, ms);
  } else if (op.a == BN_DEC_FIXED) {
    const uint32_t N = (uint32_t)op.b;                          // <= 16 (schema gate)
    if (N < 16u) {      // the value must BE an N-byte two's complement number: bytes N..15 pure sign extension of byte N-1
      const uint32_t sign = (bin_byte(v, N - 1) & 0x80u) ? 0xFFu : 0u;   // (the decoder is strict the other way: E_DECIMAL)
      bool fits = true;
      for (uint32_t j = N; j < 16u; j++) fits = fits && bin_byte(v, j) == sign;
      if (!fits) { L.err = EE_DECIMAL; L.eop = (uint32_t)N; L.edetail = c.row(op.dom); return; }
    }
    if (MODE == M_SIZE) { L.len += N; return; }
    for (uint32_t j = N; j-- > 0;) put_byte<MODE>(c, L, (uint8_t)bin_byte(v, j));
  } else if (op.a == BN_DEC_BYTES) {
    // minimal length: the magnitude bits of v (of ~v when negative) plus a sign bit, in whole bytes, at least one
    const bool neg = (v.hi >> 63) != 0;
    const uint64_t mh = neg ? ~v.hi : v.hi, ml = neg ? ~v.lo : v.lo;
    const uint32_t bits = mh ? 128u - (uint32_t)__builtin_clzll(mh) : (ml ? 64u - (uint32_t)__builtin_clzll(ml) : 0u);
    const uint32_t nb = bits / 8u + 1u;                         // 1..16
    put_varint<MODE>(c, L, (int64_t)nb);
    if (MODE == M_SIZE) { L.len += nb; return; }
    for (uint32_t j = nb; j-- > 0;) put_byte<MODE>(c, L, (uint8_t)bin_byte(v, j));
  } else {                                                       // BN_UUID_STR
    put_varint<MODE>(c, L, 36);
    if (MODE == M_SIZE) { L.len += 36; return; }
    for (uint32_t i = 0; i < 16; i++) {
      const uint32_t b = bin_byte(v, i);
      put_byte<MODE>(c, L, hex_digit(b >> 4));
      put_byte<MODE>(c, L, hex_digit(b & 15u));
      if (i == 3 || i == 5 || i == 7 || i == 9) put_byte<MODE>(c, L, (uint8_t)'-');
    }
  }
}

// ---- writes ----------------------------------------------------------------------------------------------------
// fast_encode.rs:391-399, 407-455
template <int MODE, class Ctx>
__device__ __forceinline__ void e_fixed_put(const Ctx& c, ELane& L, const Op op, const FixedV v) {
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  if (op.a == FK_I32 || op.a == FK_I64) put_varint<MODE>(c, L, (int64_t)v.bits);
  else if (op.a == FK_F32) put_raw<MODE, 4>(c, L, v.bits);
  else if (op.a == FK_F64) put_raw<MODE, 8>(c, L, v.bits);
  else put_byte<MODE>(c, L, (uint8_t)(v.bits & 1));
}

// write_string, fast_encode.rs:593-597: e_string_fetch / e_string_cofetch request the bytes, e_string_put_pf / _cf write them
// The second-level loads of a row, issued for several strings at once as soon as their spans are known: every lane
// reads 32 bytes at a valid offset (offsets of null slots are valid too; the host pads every data buffer by 64 bytes).
// The kernel is bound by the L1 (TCP) access rate, which a scattered per-lane load costs ~64-90 accesses per wave
// instruction whatever its width: no second 16-byte load when no string of the wave needs it.
template <int MODE, class Ctx>
__device__ __forceinline__ Str32 e_string_fetch(const Ctx& c, const Op op, const SpanV v) {
  Str32 d;
#pragma unroll
  for (int k = 0; k < 9; k++) d.w[k] = 0;
  if (MODE == M_SIZE) return d;
  const RH_GLOBAL uint8_t* s = reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + v.s0;
  const uint32_t n = v.s1 - v.s0;
  if (__any(n > 16u)) {
    ld_32(s, d.w);
  } else if (__any(n > 8u)) {
    const v4w a = *reinterpret_cast<const RH_GLOBAL v4wu*>(s);
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w;
  } else {
    const uint64_t a = *reinterpret_cast<const RH_GLOBAL u64u*>(s);
    d.w[0] = (uint32_t)a; d.w[1] = (uint32_t)(a >> 32);
  }
  return d;
}
// A use of both 16-byte halves AFTER the put: without it the compiler sinks the fetch into the branch that consumes
// it (a nullable string's `valid` side), i.e. back behind the wait it was hoisted to avoid.
__device__ __forceinline__ void keep_fetched(const Str32& pf) { asm volatile("" ::"v"(pf.w[0]), "v"(pf.w[4])); }

template <int MODE, class Ctx>
__device__ __forceinline__ void e_string_put_pf1(const Ctx& c, ELane& L, const Op op, const SpanV v, const Str32& pf) {
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  const uint32_t n = v.s1 - v.s0;
  put_varint<MODE>(c, L, (int64_t)n);
  put_bytes_pf<MODE>(c, L, reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + v.s0, n, pf);
}
template <int MODE, class Ctx>
__device__ __forceinline__ void e_string_put_pf(const Ctx& c, ELane& L, const Op op, const SpanV v, const Str32& pf) {
  e_string_put_pf1<MODE>(c, L, op, v, pf);
  if (MODE != M_SIZE) keep_fetched(pf);
}
// Cooperative fetch (rows of domain 0, staged emit): the strings of a wave's 64 consecutive rows are one contiguous
// byte range of the column.  When it is at most kStageBytes, lane l requests the ALIGNED 16 bytes [base + 16 l, +16)
// -- one fully coalesced load for the wave (16 L1 accesses) instead of two scattered per-lane ones (~160) -- and the
// put side transposes through the wave's LDS staging area: ds_write_b128, then every lane reads its own bytes back
// as aligned dwords + v_alignbyte.  No barrier: a wave's DS instructions execute in order.
struct StrF { Str32 pf; uint32_t base; bool coop; };

template <int MODE, class Ctx>
__device__ __forceinline__ StrF e_string_cofetch(const Ctx& c, const Op op, const SpanV v) {
  StrF f;
  f.base = 0; f.coop = false;
  if (MODE == M_STAGED && c.stage) {
    const uint32_t b = __builtin_amdgcn_readfirstlane(v.s0) & ~15u;
    const uint32_t e = __builtin_amdgcn_readlane(v.s1, 63);
    if (e - b <= kStageBytes) {                              // wave-uniform
#pragma unroll
      for (int k = 0; k < 9; k++) f.pf.w[k] = 0;
      f.coop = true; f.base = b;
      const uint32_t o = b + 16u * (threadIdx.x & 63u);
      if (o < e) {
        const v4w a = *reinterpret_cast<const RH_GLOBAL v4w*>(reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + o);
        f.pf.w[0] = a.x; f.pf.w[1] = a.y; f.pf.w[2] = a.z; f.pf.w[3] = a.w;
      }
      return f;
    }
  }
  f.pf = e_string_fetch<MODE>(c, op, v);
  return f;
}

template <int MODE, class Ctx>
__device__ __forceinline__ void e_string_put_cf(const Ctx& c, ELane& L, const Op op, const SpanV v, const StrF& f) {
  if (MODE != M_STAGED || !f.coop) {                          // wave-uniform
    e_string_put_pf<MODE>(c, L, op, v, f.pf);
    return;
  }
  RH_LDS uint8_t* st = c.stage;
  v4w a; a.x = f.pf.w[0]; a.y = f.pf.w[1]; a.z = f.pf.w[2]; a.w = f.pf.w[3];
  *reinterpret_cast<RH_LDS v4w*>(st + 16u * (threadIdx.x & 63u)) = a;        // every lane, present or not
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  const uint32_t n = v.s1 - v.s0;
  put_varint<MODE>(c, L, (int64_t)n);
  if (n > 0) {
    const uint32_t p = v.s0 - f.base;                         // <= kStageBytes; reads end before kStageStride
    const RH_LDS uint32_t* q = reinterpret_cast<const RH_LDS uint32_t*>(st + (p & ~3u));
    const uint32_t sh = p & 3u;
    uint32_t w[9];
    uint32_t prev = q[0];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t next = q[k + 1];
      w[k] = __builtin_amdgcn_alignbyte(next, prev, sh);
      prev = next;
    }
    w[8] = 0;
    RH_LDS uint8_t* d = c.lout + L.len;
    lds_put_32(d, w, n < 32u ? n : 32u);
    const RH_GLOBAL uint8_t* s = reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + v.s0;
    for (uint32_t j = 32; j < n; j += 32) {
      uint32_t x[9];
      ld_32(s + j, x);
      const uint32_t m = n - j;
      lds_put_32(d + j, x, m < 32u ? m : 32u);
    }
  }
  L.len += n;
}

// first 16 bytes of an enum's symbol text (symbols of <= 16 bytes are matched in registers)
// ... of an enum whose longest symbol has NB bytes: no wider a load than that
template <int NB, class Ctx>
__device__ __forceinline__ v4w e_enum_fetch_n(const Ctx& c, const Op op, const SpanV v) {
  const RH_GLOBAL uint8_t* s = reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + v.s0;
  v4w a;
  a.x = 0; a.y = 0; a.z = 0; a.w = 0;
  if (NB <= 4) {
    a.x = *reinterpret_cast<const RH_GLOBAL u32u*>(s);
  } else if (NB <= 8) {
    const uint64_t x = *reinterpret_cast<const RH_GLOBAL u64u*>(s);
    a.x = (uint32_t)x; a.y = (uint32_t)(x >> 32);
  } else {
    a = *reinterpret_cast<const RH_GLOBAL v4wu*>(s);
  }
  return a;
}
template <int MODE, class Ctx, class Find>
__device__ __forceinline__ void e_enum_put_pf(const Ctx& c, ELane& L, const Op op, const SpanV v, int pc, const v4w a, const Find& find) {
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  const int32_t found = find(a, v.s1 - v.s0);
  if (found < 0) { L.err = EE_ENUM; L.eop = (uint32_t)pc; L.edetail = c.row(op.dom); }
  else put_varint<MODE>(c, L, found);
}

// symbol text -> index by comparing against the schema's symbol table in HBM (any symbol length)
struct EnumTableFinder {
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  int32_t first, count;
  __device__ __forceinline__ int32_t operator()(const RH_GLOBAL uint8_t* sp, uint32_t n) const {
    int32_t found = -1;
    for (int32_t s = 0; s < count && found < 0; s++) {
      const uint32_t a = sym_off[first + s], b = sym_off[first + s + 1];
      if (b - a != n) continue;
      bool eq = true;
      for (uint32_t j = 0; j < n && eq; j++) eq = sp[j] == sym_data[a + j];
      if (eq) found = s;
    }
    return found;
  }
};

// write_enum_idx, fast_encode.rs:574-581.  find(sp, n) -> symbol index or -1.
template <int MODE, class Ctx, class Find>
__device__ __forceinline__ void e_enum_put(const Ctx& c, ELane& L, const Op op, const SpanV v, int pc, const Find& find) {
  if (!L.writes()) return;
  if (op.flags & F_NULLABLE) {
    put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (!v.valid) return;
  }
  const int32_t found = find(reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + v.s0, v.s1 - v.s0);
  if (found < 0) { L.err = EE_ENUM; L.eop = (uint32_t)pc; L.edetail = c.row(op.dom); }
  else put_varint<MODE>(c, L, found);
}

// NullableRecord, 465-476: the struct's own validity
template <int MODE, class Ctx>
__device__ __forceinline__ void e_rec_begin(const Ctx& c, ELane& L, const Op op, const bool valid) {
  const bool wr = L.writes();
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  if (wr) put_branch<MODE>(c, L, !valid, (op.flags & F_NULL_FIRST) != 0);
  L.pres = wr && valid;
}
__device__ __forceinline__ void e_rec_end(ELane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
}

// UnionEncoder::write, 507-521 (sparse: children share the row)
template <int MODE, class Ctx>
__device__ __forceinline__ void e_union_begin(const Ctx& c, ELane& L, const Op op, const int32_t t, int pc) {
  const bool wr = L.writes();
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  L.sstk = (L.sstk << 8) | 0xFFull;
  if (wr) {
    if (t < 0 || t >= op.a) { L.err = EE_UNION; L.eop = (uint32_t)pc; L.edetail = t; }
    else {
      put_varint<MODE>(c, L, t);
      L.sstk = (L.sstk & ~0xFFull) | (uint64_t)t;
    }
  }
}
__device__ __forceinline__ void e_variant(ELane& L, const Op op) {
  L.pres = (L.pstk & 1) && ((uint32_t)(L.sstk & 0xFF) == (uint32_t)op.a);
}
__device__ __forceinline__ void e_union_end(ELane& L) {
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
  L.sstk >>= 8;
}

// ListEncoder / MapEncoder::write, 525-561 (+ Nullable*, 478-496)
template <int MODE, class Ctx>
__device__ __forceinline__ void e_list_begin(Ctx& c, ELane& L, const Op op, const SpanV v) {
  const bool wr = L.writes();
  L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
  L.lstk = (L.lstk << 1) | (L.live ? 1u : 0u);
  uint32_t n = 0;
  bool has = false;
  if (wr) {
    if (op.flags & F_NULLABLE) put_branch<MODE>(c, L, !v.valid, (op.flags & F_NULL_FIRST) != 0);
    if (v.valid) {
      n = v.s1 - v.s0;
      if (n > 0) put_varint<MODE>(c, L, (int64_t)n);
      c.set_row(op.a, v.s0);            // op.a = child row domain: first item
      has = true;
    }
  }
  c.set_rem(op.c, n);
  // lstk bit 0 remembers "this lane owes a 0 terminator"; the saved `live` sits one bit above it
  L.lstk = (L.lstk << 1) | (has ? 1u : 0u);
  L.live = has;
  L.pres = has;
}
template <class Ctx>
__device__ __forceinline__ bool e_list_next(const Ctx& c, ELane& L, const Op op) {   // false: no lane has an item left
  const bool item = L.live && L.err == 0 && c.rem(op.c) > 0;
  L.pres = item;
  return __any(item);
}
template <class Ctx>
__device__ __forceinline__ void e_list_tail(Ctx& c, ELane& L, const Op op) {
  const uint32_t left = c.rem(op.c);
  if (L.live && L.err == 0 && left > 0) {
    c.set_rem(op.c, left - 1);
    c.set_row(op.a, c.row(op.a) + 1);
  }
}
template <int MODE, class Ctx>
__device__ __forceinline__ void e_list_end(const Ctx& c, ELane& L) {
  const bool owes = (L.lstk & 1) != 0;
  L.lstk >>= 1;
  L.live = L.lstk & 1;
  L.lstk >>= 1;
  L.pres = L.pstk & 1;
  L.pstk >>= 1;
  if (owes && L.live && L.err == 0) put_byte<MODE>(c, L, 0);    // terminator (an empty list is just this 0)
}

// ---- workgroup frame -------------------------------------------------------------------------------------------
__device__ __forceinline__ Geo egeometry(const EParams& P, uint32_t b) {
  Geo g;
  uint32_t chunk = b / P.bpc;
  if (chunk > P.k - 1) chunk = P.k - 1;
  const uint32_t lb = b - chunk * P.bpc;
  const uint64_t rows_c = chunk == P.k - 1 ? P.rows_last : P.sz;
  g.chunk = chunk;
  g.lrow0 = lb * kBlock;
  g.rec0 = (uint64_t)chunk * P.sz + g.lrow0;
  const uint64_t left = rows_c - g.lrow0;
  g.nrec = left < (uint64_t)kBlock ? (uint32_t)left : (uint32_t)kBlock;
  return g;
}

__device__ __forceinline__ void elane_init(ELane& L, const Geo& g, uint32_t tid) {
  L.len = 0; L.err = 0; L.edetail = 0; L.eop = 0;
  L.live = tid < g.nrec; L.pres = L.live;
  L.pstk = 0; L.lstk = 0; L.sstk = 0;
}

// misc[0] lowest erroring tid, misc[4..7] wave totals
__device__ __forceinline__ void ereport(const EParams& P, uint32_t* misc, const ELane& L, const Geo& g, uint32_t tid) {
  if (L.err) atomicMin(&misc[0], tid);
  __syncthreads();
  if (misc[0] == tid) {
    ErrInfo ei; ei.code = L.err; ei.pad = L.eop; ei.detail = L.edetail;
    P.errinfo[blockIdx.x] = ei;
    atomicMax(P.first_bad, ~(unsigned long long)(g.rec0 + tid));
  }
}

// W: the walker.  W::Ctx is its lane context, W::kCursorWords(P) the LDS words its row cursors take in front of
// misc[8] (0 when they live in registers), W::walk<MODE>(c, L) the per-row walk.
//
//   size: walk 1, the encoded length of every row -> rowlen[], per-workgroup sums, first failing row
template <class W>
__device__ __forceinline__ void e_size_body(const EParams& P, uint8_t* smem) {
  uint32_t* cursors = reinterpret_cast<uint32_t*>(smem);
  uint32_t* misc = cursors + W::cursor_words(P);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = egeometry(P, blockIdx.x);
  if (tid == 0) misc[0] = 0xFFFFFFFFu;
  ELane L;
  elane_init(L, g, tid);
  typename W::Ctx c;
  W::init(c, P, cursors, g, tid);
  __syncthreads();
  W::template walk<M_SIZE>(c, L);
  const uint32_t mylen = tid < g.nrec ? L.len : 0u;
  P.rowlen[(size_t)blockIdx.x * kBlock + tid] = mylen;
  const uint32_t v = wave_sum(mylen);
  if (lane == 0) misc[4 + wave] = v;
  ereport(P, misc, L, g, tid);     // barrier inside
  if (tid == 0) P.blocksum[blockIdx.x] = misc[4] + misc[5] + misc[6] + misc[7];
}

//   emit: row lengths back from the size pass, scan inside the workgroup, offsets[row+1], then walk 2 writes the
//   datum bytes -- into an LDS window laid out congruent (mod 16) to the workgroup's contiguous output range, which
//   the whole workgroup then streams to HBM with aligned 16-byte stores.  A workgroup whose rows do not fit the
//   window stores straight to HBM (per-lane byte stores).
template <class W>
__device__ __forceinline__ void e_emit_body(const EParams& P, uint8_t* smem) {
  uint32_t* cursors = reinterpret_cast<uint32_t*>(smem);
  uint32_t* misc = cursors + W::cursor_words(P);
  uint8_t* const window = reinterpret_cast<uint8_t*>(misc + 8);        // 16-byte aligned
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = egeometry(P, blockIdx.x);
  ELane L;
  L.clk.init();
  elane_init(L, g, tid);
  typename W::Ctx c;
  W::init(c, P, cursors, g, tid);
  const uint32_t mylen = P.rowlen[(size_t)blockIdx.x * kBlock + tid];   // 0 beyond the chunk's rows
  const uint32_t incl = wave_incl_scan(mylen, lane);
  if (lane == 63) misc[4 + wave] = incl;
  __syncthreads();
  L.clk.mark(0);
  const uint32_t base = P.blockbase[blockIdx.x];                        // chunk-relative first byte of this workgroup
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; w++) {
    const uint32_t t = misc[4 + w];
    if (w < wave) before += t;
    total += t;
  }
  const uint32_t rel = before + incl - mylen;                           // first byte of this row inside the workgroup's range
  RH_GLOBAL int32_t* offs = reinterpret_cast<RH_GLOBAL int32_t*>(reinterpret_cast<uintptr_t>(P.outptr[(size_t)g.chunk * 2]));
  RH_GLOBAL uint8_t* data = reinterpret_cast<RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(P.outptr[(size_t)g.chunk * 2 + 1]));
  if (tid < g.nrec) offs[g.lrow0 + tid + 1] = (int32_t)(base + rel + mylen);
  if (g.lrow0 == 0 && tid == 0) offs[0] = 0;

  const uint32_t shift = base & 15u;          // data is 256-byte aligned: LDS window offset == HBM address (mod 16)
  c.out = data + base + rel;
  c.lout = (RH_LDS uint8_t*)(window + shift + rel);
  c.stage = P.stage_bytes ? (RH_LDS uint8_t*)(window + P.win_bytes + wave * kStageStride) : (RH_LDS uint8_t*)nullptr;
  if (shift + total > P.win_bytes) {          // uniform: this workgroup's rows do not fit the window
    W::template walk<M_DIRECT>(c, L);
    return;
  }
  L.clk.mark(1);
  W::template walk<M_STAGED>(c, L);
  L.clk.mark(20);
  __syncthreads();
  L.clk.mark(21);
  // window[shift, shift+total) -> data[base, base+total): byte head up to the first 16-byte boundary, aligned
  // 16-byte body (ds_read_b128 -> global_store_dwordx4, consecutive lanes = consecutive lines), byte tail
  RH_GLOBAL uint8_t* dst = data + base;
  uint32_t head = (16u - shift) & 15u;
  if (head > total) head = total;
  if (tid < head) dst[tid] = window[shift + tid];
  const uint32_t nvec = (total - head) >> 4;
  const v4u* src16 = reinterpret_cast<const v4u*>(window + shift + head);
  RH_GLOBAL v4u* dst16 = reinterpret_cast<RH_GLOBAL v4u*>(dst + head);
  for (uint32_t i = tid; i < nvec; i += kBlock) dst16[i] = src16[i];
  const uint32_t done = head + (nvec << 4);
  if (tid < total - done) dst[done + tid] = window[shift + done + tid];
  L.clk.mark(22);
  L.clk.flush(P.prof);
}

// LDS bytes in front of rh_e_emit's staging window: cursor words of the walker + misc[8]
__host__ __device__ inline uint32_t enc_lds_fixed_bytes(uint32_t cursor_words) { return (cursor_words + 8) * 4; }

}  // namespace rh
