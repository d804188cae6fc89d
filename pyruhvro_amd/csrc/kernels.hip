// CDNA4 (gfx950) kernels of the Avro -> Arrow direct-decode path.
//
// One workgroup = 256 consecutive records of one output chunk, one lane per
// record (the per-record unit of work of the reference's hot loop,
// ruhvro/src/fast_decode.rs:825-828).  The workgroup's input bytes are one
// contiguous window of the packed payload, staged into LDS with 16-byte
// coalesced loads; every lane then walks its own record out of LDS.
//
// The walker is a WAVE-UNIFORM interpreter of the schema program
// (program.h): the program counter is scalar, all 64 lanes execute the same
// op, and the reference's data-dependent control flow becomes per-lane
// predicates:
//   * live  -- the lane owns a row in the current row domain,
//   * pres  -- the row is decoded from bytes (true) or null-filled (false),
//              i.e. FieldDecoder::decode vs FieldDecoder::append_null
//              (fast_decode.rs:421-499 vs 503-534).
// Nullable records and N-variant unions only flip `pres` (children of a null
// record / non-selected variants are visited in null-fill mode, exactly the
// sparse-union / null-struct fill of fast_decode.rs:608-616,649-655); array
// and map blocks run as a wave loop that lasts as long as any lane still has
// items (ballot), lanes without an item are simply not live.
//
// Kernels:
//   k_size  walk 1: per-record counters (child rows per array/map domain,
//           bytes per string column) -> per-workgroup sums; malformed input
//           is detected here (lowest failing record wins).
//   k_scan  chunk-segmented exclusive scan of the workgroup sums -> each
//           workgroup's base row / base byte per counter, and chunk totals
//           (= Arrow buffer sizes).
//   k_init  offsets[0] = 0, zero the bitmaps written with atomics.
//   k_emit  recomputes the counters of its 256 records, scans them inside the
//           workgroup, then walk 2 writes every Arrow buffer: values and
//           offsets at [row] (coalesced), validity / boolean bitmaps with one
//           64-bit ballot store per wavefront, string bytes at the scanned
//           byte offsets, sparse-union type ids.
// HBM-bound byte shuffling: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.h"

namespace rh {

// --------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
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
// per-lane walker state
// --------------------------------------------------------------------------
struct Lane {
  uint32_t cur, end;   // byte cursor / record end, relative to the window base
  uint32_t err;        // ErrCode, 0 = ok
  int64_t edetail;
  bool live, pres;
  uint32_t pstk;       // saved `pres` bits   (nullable record / union / list)
  uint32_t lstk;       // saved `live` bits   (list)
  uint64_t sstk;       // saved union selectors, 8 bits each
};

struct Ctx {
  const KParams* P;
  uint32_t* cnt;       // LDS [K][256]   running child row / byte offset per counter
  uint32_t* rem;       // LDS [depth][256] items left in the current block
  uint32_t* nullcnt;   // LDS [nnodes]
  uint32_t chunk;
  uint32_t lrow;       // chunk-local row of this lane (domain 0)
  uint32_t tid, lane;
  bool wave_live;      // the wave owns at least one row (uniform)
};

// zig-zag LEB128 varint, fast_decode.rs:854-869
template <typename SrcT>
__device__ __forceinline__ uint32_t rd_varint(SrcT src, uint32_t& cur, uint32_t end, int64_t& out) {
  uint64_t r = 0;
  uint32_t shift = 0;
  for (;;) {
    if (cur >= end) return E_EOB;
    uint32_t b = src[cur++];
    r |= (uint64_t)(b & 0x7F) << shift;
    if ((b & 0x80) == 0) break;
    shift += 7;
    if (shift >= 64) return E_VARINT;
  }
  out = (int64_t)(r >> 1) ^ -(int64_t)(r & 1);
  return E_OK;
}

// union_branch, fast_decode.rs:585-593.  Returns true for the value branch.
template <typename SrcT>
__device__ __forceinline__ bool rd_branch(SrcT src, Lane& L, bool null_first) {
  int64_t idx = 0;
  uint32_t e = rd_varint(src, L.cur, L.end, idx);
  if (e) { L.err = e; return false; }
  if (idx == 0) return !null_first;
  if (idx == 1) return null_first;
  L.err = E_BRANCH;
  L.edetail = idx;
  return false;
}

__device__ __forceinline__ void* bufp(const Ctx& c, int32_t id) {
  return c.P->bufptr[(size_t)id * c.P->k + c.chunk];
}

__device__ __forceinline__ uint32_t row_of(const Ctx& c, const Op& op) {
  return op.dom == 0 ? c.lrow : c.cnt[(op.dom - 1) * kBlock + c.tid];
}

// validity bit + null count of one row (buffer exists iff F_CAN_NULL)
template <bool EMIT>
__device__ __forceinline__ void put_validity(const Ctx& c, const Op& op, bool act, bool valid, uint32_t row) {
  if (!EMIT) return;
  if (!(op.flags & F_CAN_NULL)) return;
  if (op.dom == 0) {
    uint64_t m = __ballot(valid);
    uint64_t nm = __ballot(act && !valid);
    if (c.lane == 0 && c.wave_live) {
      reinterpret_cast<uint64_t*>(bufp(c, op.buf0))[c.lrow >> 6] = m;
      if (nm) atomicAdd(&c.nullcnt[op.node], (uint32_t)__popcll(nm));
    }
  } else if (act) {
    if (valid) atomicOr(&reinterpret_cast<uint32_t*>(bufp(c, op.buf0))[row >> 5], 1u << (row & 31));
    else atomicAdd(&c.nullcnt[op.node], 1u);
  }
}

// --------------------------------------------------------------------------
// the interpreter
// --------------------------------------------------------------------------
template <bool EMIT, typename SrcT>
__device__ __forceinline__ void walk(const Ctx& c, SrcT src, Lane& L) {
  const KParams& P = *c.P;
  int pc = 0;
  for (;;) {
    pc = __builtin_amdgcn_readfirstlane(pc);
    const Op op = P.prog[pc];
    const bool act = L.live && L.err == 0;
    switch (op.code) {
      case OP_END:
        return;

      case OP_FIXED: {
        const bool dec = act && L.pres;
        bool isval = dec;
        if ((op.flags & F_NULLABLE) && dec) isval = rd_branch(src, L, op.flags & F_NULL_FIRST);
        uint64_t bits = 0;
        if (dec && isval && L.err == 0) {
          if (op.a == FK_I32 || op.a == FK_I64) {
            int64_t v = 0;
            uint32_t e = rd_varint(src, L.cur, L.end, v);
            if (e) L.err = e;
            bits = op.a == FK_I32 ? (uint64_t)(uint32_t)(int32_t)v : (uint64_t)v;   // `as i32` truncates
          } else if (op.a == FK_F32) {
            if (L.end - L.cur < 4) L.err = E_EOB_F32;
            else {
              for (int j = 0; j < 4; j++) bits |= (uint64_t)src[L.cur + j] << (8 * j);
              L.cur += 4;
            }
          } else if (op.a == FK_F64) {
            if (L.end - L.cur < 8) L.err = E_EOB_F64;
            else {
              for (int j = 0; j < 8; j++) bits |= (uint64_t)src[L.cur + j] << (8 * j);
              L.cur += 8;
            }
          } else {  // FK_BOOL, fast_decode.rs:893-900
            if (L.cur >= L.end) L.err = E_EOB;
            else {
              uint32_t b = src[L.cur++];
              if (b > 1) { L.err = E_BOOL; L.edetail = b; }
              bits = b;
            }
          }
        }
        const bool valid = dec && isval && L.err == 0;
        if (!valid) bits = 0;   // zero under nulls (arrow-rs append_null)
        uint32_t row = 0;
        if (EMIT) {
          row = row_of(c, op);
          if (op.a == FK_BOOL) {
            if (op.dom == 0) {
              uint64_t m = __ballot(bits != 0);
              if (c.lane == 0 && c.wave_live) reinterpret_cast<uint64_t*>(bufp(c, op.buf1))[c.lrow >> 6] = m;
            } else if (act && bits) {
              atomicOr(&reinterpret_cast<uint32_t*>(bufp(c, op.buf1))[row >> 5], 1u << (row & 31));
            }
          } else if (act) {
            if (op.a == FK_I32 || op.a == FK_F32) reinterpret_cast<uint32_t*>(bufp(c, op.buf1))[row] = (uint32_t)bits;
            else reinterpret_cast<uint64_t*>(bufp(c, op.buf1))[row] = bits;
          }
        }
        put_validity<EMIT>(c, op, act, valid, row);
        break;
      }

      case OP_STRING:
      case OP_ENUM: {
        const bool dec = act && L.pres;
        bool isval = dec;
        if ((op.flags & F_NULLABLE) && dec) isval = rd_branch(src, L, op.flags & F_NULL_FIRST);
        uint32_t len = 0, spos = 0;
        if (dec && isval && L.err == 0) {
          int64_t v = 0;
          uint32_t e = rd_varint(src, L.cur, L.end, v);
          if (e) L.err = e;
          else if (op.code == OP_STRING) {          // read_string, fast_decode.rs:902-922
            if (v < 0) L.err = E_NEGLEN;
            else if ((uint64_t)(L.end - L.cur) < (uint64_t)v) L.err = E_EOB_STR;
            else { len = (uint32_t)v; spos = L.cur; L.cur += len; }
          } else {                                  // append_enum, fast_decode.rs:570-578
            if ((uint64_t)v >= (uint64_t)op.c) { L.err = E_ENUM; L.edetail = v; }
            else {
              spos = P.sym_off[op.b + (int32_t)v];
              len = P.sym_off[op.b + (int32_t)v + 1] - spos;
            }
          }
        }
        const bool valid = dec && isval && L.err == 0;
        if (!valid) len = 0;
        uint32_t* bo = &c.cnt[op.a * kBlock + c.tid];
        const uint32_t o = *bo;
        uint32_t row = 0;
        if (EMIT) {
          row = row_of(c, op);
          if (act) {
            reinterpret_cast<uint32_t*>(bufp(c, op.buf1))[row + 1] = o + len;   // offsets repeat under nulls
            if (len) {
              uint8_t* d = reinterpret_cast<uint8_t*>(bufp(c, op.buf2)) + o;
              if (op.code == OP_STRING) {
                for (uint32_t j = 0; j < len; j++) d[j] = src[spos + j];
              } else {
                for (uint32_t j = 0; j < len; j++) d[j] = P.sym_data[spos + j];
              }
            }
          }
        }
        if (act) *bo = o + len;
        put_validity<EMIT>(c, op, act, valid, row);
        break;
      }

      case OP_REC_BEGIN: {   // NullableRecord, fast_decode.rs:482-485 + 595-616
        L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
        const bool dec = act && L.pres;
        bool isval = dec;
        if (dec) isval = rd_branch(src, L, op.flags & F_NULL_FIRST);
        const bool valid = dec && isval && L.err == 0;
        put_validity<EMIT>(c, op, act, valid, EMIT ? row_of(c, op) : 0);
        L.pres = valid;      // null record -> children are null-filled
        break;
      }
      case OP_REC_END:
        L.pres = L.pstk & 1;
        L.pstk >>= 1;
        break;

      case OP_UNION_BEGIN: {   // UnionDecoder::decode / append_null, fast_decode.rs:643-668
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
        if (EMIT && act) reinterpret_cast<int8_t*>(bufp(c, op.buf1))[row_of(c, op)] = (int8_t)tidv;
        break;
      }
      case OP_VARIANT:
        L.pres = (L.pstk & 1) && ((uint32_t)(L.sstk & 0xFF) == (uint32_t)op.a);
        break;
      case OP_UNION_END:
        L.pres = L.pstk & 1;
        L.pstk >>= 1;
        L.sstk >>= 8;
        break;

      case OP_LIST_BEGIN: {   // ListDecoder / MapDecoder (+ Nullable*), fast_decode.rs:487-496,703-770
        L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
        L.lstk = (L.lstk << 1) | (L.live ? 1u : 0u);
        const bool dec = act && L.pres;
        bool isval = dec;
        if ((op.flags & F_NULLABLE) && dec) isval = rd_branch(src, L, op.flags & F_NULL_FIRST);
        const bool valid = dec && isval && L.err == 0;
        put_validity<EMIT>(c, op, act, valid, EMIT ? row_of(c, op) : 0);
        L.live = valid;      // only rows that really carry a list enter the block loop
        L.pres = valid;
        c.rem[op.c * kBlock + c.tid] = 0;
        break;
      }
      case OP_LIST_NEXT: {    // read_block_count, fast_decode.rs:689-700
        uint32_t* rm = &c.rem[op.c * kBlock + c.tid];
        if (act && *rm == 0) {
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
            // Clamp the trip count: with m = min wire bytes per item and R bytes left, no more
            // than R/m items can decode, so item R/m+1 raises the same error the reference hits.
            const uint64_t R = L.end - L.cur;
            if (op.buf2 /*min wire bytes per item*/ > 0) {
              uint64_t cap = R / (uint32_t)op.buf2 + 1;
              *rm = (uint32_t)((uint64_t)n < cap ? (uint64_t)n : cap);
            } else if ((uint64_t)n > 0x00FFFFFFull) {
              L.err = E_LIST_RANGE; L.edetail = n;
            } else {
              *rm = (uint32_t)n;
            }
            break;
          }
        }
        const bool item = L.live && L.err == 0;
        if (!__any(item)) { pc = op.b; continue; }
        L.pres = L.live;
        break;
      }
      case OP_LIST_TAIL: {
        if (act) {
          c.rem[op.c * kBlock + c.tid] -= 1;
          c.cnt[(op.a - 1) * kBlock + c.tid] += 1;     // op.a = child row domain
        }
        pc = op.b;
        continue;
      }
      case OP_LIST_END: {
        L.live = L.lstk & 1;
        L.lstk >>= 1;
        L.pres = L.pstk & 1;
        L.pstk >>= 1;
        if (EMIT && L.live && L.err == 0) {
          // cumulative child rows so far == Arrow offset of the next row (null / empty rows repeat it)
          reinterpret_cast<uint32_t*>(bufp(c, op.buf1))[row_of(c, op) + 1] = c.cnt[(op.a - 1) * kBlock + c.tid];
        }
        break;
      }
      default:
        return;
    }
    pc++;
  }
}

// --------------------------------------------------------------------------
// workgroup geometry + LDS carving shared by k_size / k_emit
// --------------------------------------------------------------------------
struct Geo {
  uint32_t chunk;
  uint32_t lrow0;     // chunk-local row of lane 0 of the workgroup
  uint64_t rec0;      // global record index of lane 0
  uint32_t nrec;      // live rows in this workgroup (1..256)
};

__device__ __forceinline__ Geo geometry(const KParams& P, uint32_t b) {
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

struct Smem {
  uint32_t* cnt;      // [K][256]
  uint32_t* rem;      // [list_depth][256]
  uint32_t* nullcnt;  // [nnodes]
  uint32_t* wtot;     // [K][4]
  uint32_t* misc;     // [4]: 0 = lowest erroring tid
  uint8_t* win;       // input window (16-byte aligned)
};

__device__ __forceinline__ Smem carve(const KParams& P, uint8_t* smem) {
  Smem s;
  uint32_t* p = reinterpret_cast<uint32_t*>(smem);
  s.cnt = p; p += (P.K > 0 ? P.K : 1) * kBlock;
  s.rem = p; p += (P.list_depth > 0 ? P.list_depth : 1) * kBlock;
  s.wtot = p; p += (P.K > 0 ? P.K : 1) * 4;
  s.nullcnt = p; p += ((P.nnodes + 3) & ~3);
  s.misc = p; p += 4;
  s.win = reinterpret_cast<uint8_t*>(p);   // all pieces above are multiples of 16 bytes
  return s;
}

// Host mirror of carve(): LDS bytes in front of the window.
extern "C" uint32_t rh_lds_fixed_bytes(int K, int list_depth, int nnodes) {
  uint32_t w = (uint32_t)(K > 0 ? K : 1) * kBlock + (uint32_t)(list_depth > 0 ? list_depth : 1) * kBlock +
               (uint32_t)(K > 0 ? K : 1) * 4 + (uint32_t)((nnodes + 3) & ~3) + 4;
  return w * 4;
}

// Stage [wb16, we) of the payload into LDS with 16-byte loads.
__device__ __forceinline__ void stage_window(const KParams& P, uint8_t* win, uint64_t wb16, uint64_t we, uint32_t tid) {
  const uint32_t nvec = (uint32_t)((we - wb16 + 15) >> 4);
  const uint8_t* g = P.data + wb16;
  for (uint32_t i = tid; i < nvec; i += kBlock) {
    const uint64_t pos = wb16 + ((uint64_t)i << 4);
    if (pos + 16 <= P.data_len) {
      reinterpret_cast<uint4*>(win)[i] = *reinterpret_cast<const uint4*>(g + ((size_t)i << 4));
    } else {
      for (uint32_t j = 0; j < 16; j++) win[(i << 4) + j] = pos + j < P.data_len ? g[((size_t)i << 4) + j] : 0;
    }
  }
}

__device__ __forceinline__ void lane_init(Lane& L, const KParams& P, const Geo& g, uint64_t wb16, uint32_t tid) {
  L.live = tid < g.nrec;
  L.pres = L.live;
  L.err = 0;
  L.edetail = 0;
  L.pstk = 0; L.lstk = 0; L.sstk = 0;
  L.cur = 0; L.end = 0;
  if (L.live) {
    const uint64_t o0 = P.offsets[g.rec0 + tid], o1 = P.offsets[g.rec0 + tid + 1];
    L.cur = (uint32_t)(o0 - wb16);
    L.end = (uint32_t)(o1 - wb16);
  }
}

// Lowest erroring lane of the workgroup reports (code, detail); lowest record index wins globally
// (== the in-order join of deserialize.rs:115-119 + first `?` in fast_decode.rs:827).
__device__ __forceinline__ void report_errors(const KParams& P, const Smem& s, const Lane& L, const Geo& g, uint32_t tid) {
  if (L.err) atomicMin(&s.misc[0], tid);
  __syncthreads();
  if (s.misc[0] == tid) {
    ErrInfo ei; ei.code = L.err; ei.pad = 0; ei.detail = L.edetail;
    P.errinfo[blockIdx.x] = ei;
    atomicMax(P.first_bad, ~(unsigned long long)(g.rec0 + tid));
  }
}

template <bool EMIT>
__device__ __forceinline__ void run_walk(const Ctx& c, const Smem& s, Lane& L, bool fits, uint64_t wb16) {
  if (fits) walk<EMIT, const uint8_t*>(c, s.win, L);
  else walk<EMIT, const uint8_t*>(c, c.P->data + wb16, L);
}

// --------------------------------------------------------------------------
// k_size
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_size(KParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const Smem s = carve(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = geometry(P, blockIdx.x);
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (fits) stage_window(P, s.win, wb16, we, tid);
  for (int k = 0; k < P.K; k++) s.cnt[k * kBlock + tid] = 0;
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  __syncthreads();

  Lane L;
  lane_init(L, P, g, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) { L.err = E_EOB; L.live = true; }   // window beyond 32-bit cursors
  Ctx c;
  c.P = &P; c.cnt = s.cnt; c.rem = s.rem; c.nullcnt = s.nullcnt; c.chunk = g.chunk;
  c.lrow = g.lrow0 + tid; c.tid = tid; c.lane = lane; c.wave_live = (wave * 64) < g.nrec;
  run_walk<false>(c, s, L, fits, wb16);

  for (int k = 0; k < P.K; k++) {
    uint32_t v = wave_sum(s.cnt[k * kBlock + tid]);
    if (lane == 0) s.wtot[k * 4 + wave] = v;
  }
  report_errors(P, s, L, g, tid);   // contains the barrier that publishes wtot
  if ((int)tid < P.K)
    P.blocksum[(size_t)tid * P.nblocks + blockIdx.x] =
        s.wtot[tid * 4] + s.wtot[tid * 4 + 1] + s.wtot[tid * 4 + 2] + s.wtot[tid * 4 + 3];
}

// --------------------------------------------------------------------------
// k_scan: one workgroup per (counter, chunk)
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_scan(KParams P) {
  __shared__ uint32_t wt[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t kk = blockIdx.x % (uint32_t)P.K, ch = blockIdx.x / (uint32_t)P.K;
  const uint32_t b0 = ch * P.bpc;
  const uint32_t b1 = ch == P.k - 1 ? P.nblocks : (ch + 1) * P.bpc;
  const uint32_t* in = P.blocksum + (size_t)kk * P.nblocks;
  uint32_t* out = P.blockbase + (size_t)kk * P.nblocks;
  uint64_t carry = 0;
  for (uint32_t base = b0; base < b1; base += kBlock) {
    const uint32_t i = base + tid;
    const uint32_t v = i < b1 ? in[i] : 0;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) wt[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wt[w];
    const uint32_t tile = wt[0] + wt[1] + wt[2] + wt[3];
    if (i < b1) out[i] = (uint32_t)(carry + woff + (incl - v));
    carry += tile;
    __syncthreads();
  }
  if (tid == 0) P.totals[(size_t)kk * P.k + ch] = carry;
}

// --------------------------------------------------------------------------
// k_init: offsets[0] = 0 for every offsets buffer, zero the atomically-built bitmaps.
// grid = nbuf * k workgroups; desc/sizes live next to bufptr.
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_init(void* const* bufptr, const uint64_t* bufsize,
                                                              const BufDesc* desc, uint32_t nbuf, uint32_t k) {
  const uint32_t id = blockIdx.x / k;
  const BufDesc d = desc[id];
  uint8_t* p = reinterpret_cast<uint8_t*>(bufptr[blockIdx.x]);
  const uint64_t sz = bufsize[blockIdx.x];
  if (d.kind == BK_OFFSETS) {
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(p) = 0;
  } else if (d.kind == BK_BITMAP && d.dom != 0) {
    for (uint64_t i = threadIdx.x; i < sz / 4; i += kBlock) reinterpret_cast<uint32_t*>(p)[i] = 0;
  }
}

// --------------------------------------------------------------------------
// k_emit
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_emit(KParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const Smem s = carve(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = geometry(P, blockIdx.x);
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (fits) stage_window(P, s.win, wb16, we, tid);
  for (int k = 0; k < P.K; k++) s.cnt[k * kBlock + tid] = 0;
  for (int i = tid; i < P.nnodes; i += kBlock) s.nullcnt[i] = 0;
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  __syncthreads();

  Lane L;
  Ctx c;
  c.P = &P; c.cnt = s.cnt; c.rem = s.rem; c.nullcnt = s.nullcnt; c.chunk = g.chunk;
  c.lrow = g.lrow0 + tid; c.tid = tid; c.lane = lane; c.wave_live = (wave * 64) < g.nrec;

  if (P.K > 0) {
    // walk 1 again (cheaper than 4*K bytes/record of HBM round trip), then the in-workgroup scan
    lane_init(L, P, g, wb16, tid);
    run_walk<false>(c, s, L, fits, wb16);
    for (int k = 0; k < P.K; k++) {
      const uint32_t v = s.cnt[k * kBlock + tid];
      const uint32_t incl = wave_incl_scan(v, lane);
      if (lane == 63) s.wtot[k * 4 + wave] = incl;
      s.cnt[k * kBlock + tid] = incl - v;
    }
    __syncthreads();
    for (int k = 0; k < P.K; k++) {
      uint32_t base = P.blockbase[(size_t)k * P.nblocks + blockIdx.x];
      for (uint32_t w = 0; w < wave; w++) base += s.wtot[k * 4 + w];
      s.cnt[k * kBlock + tid] += base;
    }
  }

  lane_init(L, P, g, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;
  run_walk<true>(c, s, L, fits, wb16);

  report_errors(P, s, L, g, tid);   // barrier inside: nullcnt complete
  for (int i = tid; i < P.nnodes; i += kBlock) {
    const uint32_t v = s.nullcnt[i];
    if (v) atomicAdd(&P.nullcount[(size_t)i * P.k + g.chunk], v);
  }
}

}  // namespace rh

// --------------------------------------------------------------------------
// launchers (called from engine.cpp; plain C linkage, HIP types stay in here)
// --------------------------------------------------------------------------
extern "C" int rh_launch_size(const rh::KParams* P, uint32_t lds_bytes, void* stream) {
  hipLaunchKernelGGL(rh::rh_k_size, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_scan(const rh::KParams* P, void* stream) {
  hipLaunchKernelGGL(rh::rh_k_scan, dim3((uint32_t)P->K * P->k), dim3(rh::kBlock), 0, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_init(void* const* bufptr, const uint64_t* bufsize, const rh::BufDesc* desc, uint32_t nbuf,
                              uint32_t k, void* stream) {
  hipLaunchKernelGGL(rh::rh_k_init, dim3(nbuf * k), dim3(rh::kBlock), 0, (hipStream_t)stream, bufptr, bufsize, desc,
                     nbuf, k);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_emit(const rh::KParams* P, uint32_t lds_bytes, void* stream) {
  hipLaunchKernelGGL(rh::rh_k_emit, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_set_max_lds(uint32_t bytes) {
  int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(rh::rh_k_size), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e) return e;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(rh::rh_k_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
