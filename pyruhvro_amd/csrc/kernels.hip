// CDNA4 (gfx950) kernels of the Avro -> Arrow direct-decode path: the generic
// (any supported schema) form, driven by the schema program of program.h.
//
// One workgroup = 256 consecutive records of one output chunk, one lane per
// record.  The workgroup's input bytes are one contiguous window of the packed
// payload, staged into LDS with 16-byte coalesced loads; every lane then walks
// its own record out of LDS with unaligned 8-byte DS reads (walk.h).
//
// Kernels:
//   k_size  walk 1: per-record counters (child rows per array/map domain,
//           bytes per string column) -> per-workgroup sums; malformed input
//           is detected here (lowest failing record wins).
//   k_scan  chunk-segmented exclusive scan of the workgroup sums -> each
//           workgroup's base row / base byte per counter, and chunk totals
//           (= Arrow buffer sizes).
//   k_init  offsets[0] = 0, zero the bitmaps written with atomics.
//   k_emit  takes the counters of its 256 records as k_size left them, scans them inside the
//           workgroup, then walk 2 writes every Arrow buffer: values and
//           offsets at [row] (coalesced), validity / boolean bitmaps with one
//           64-bit ballot store per wavefront, sparse-union type ids, string
//           bytes with per-lane 8-byte stores at the scanned byte offsets.
// HBM-bound byte shuffling: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define RH_DEEP 1      // the interpreter takes any nesting the schema front-end accepts (walk.h stk_t / sel_t)

#include "kernel_common.h"
#include "program.h"
#include "walk.h"

namespace rh {

// --------------------------------------------------------------------------
// interpreter context: per-lane counters live in LDS ([id][256], conflict-free)
// --------------------------------------------------------------------------
template <int T>
struct ICtx {
  static constexpr bool kWide = true;   // 64-bit buffer indexing: any chunk size
  static constexpr bool kSkip = false;
  static constexpr bool kEnumImm = false;
  static constexpr bool kWaveCtr = true;      // (wide schemas, program.h F_WAVE_CTR: tested per op)
  static __device__ __forceinline__ bool enum_sym(int, uint32_t, uint32_t&, uint64_t&) { return false; }
  uint32_t* cnt;             // LDS [KL][T]
  uint32_t* rem;             // LDS [depth][T]
  uint32_t* nullcnt;         // LDS [nnodes]
  const uint64_t* bufs;      // LDS [nbuf]   this chunk's buffer addresses
  const uint32_t* gb;        // LDS [K]      chunk-relative base of this workgroup per counter
  uint32_t* wv;              // LDS [K]      wide schemas: the wavefront's sum (size walk) / running base (emit walk) of every wave counter
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  uint32_t lrow;             // chunk-local row of this lane (domain 0)
  uint32_t tid, lane;
  bool wave_live;            // the wave owns at least one row (uniform)

  __device__ __forceinline__ uint32_t& counter(int id) const { return cnt[id * T + tid]; }
  __device__ __forceinline__ uint32_t& remaining(int d) const { return rem[d * T + tid]; }
  __device__ __forceinline__ void* buf(int id) const { return reinterpret_cast<void*>(bufs[id]); }
  __device__ __forceinline__ uint32_t gbase(int id) const { return gb[id]; }
  __device__ __forceinline__ void wave_total(int id, uint32_t len) const {
    const uint32_t t = wave_sum(len);
    if (lane == 0) wv[id] = t;
  }
  __device__ __forceinline__ uint32_t wave_offset(int id, uint32_t len) const {
    const uint32_t incl = wave_incl_scan(len, lane);
    const uint32_t base = wv[id];
    if (lane == 63) wv[id] = base + incl;     // (DS operations of a wavefront execute in order: every lane has read `base`)
    return base + incl - len;
  }
  template <bool ACC>
  __device__ __forceinline__ void add_nulls_wave(int node, uint32_t n) const {
    if (lane == 0 && n) atomicAdd(&nullcnt[node], n);
  }
  __device__ __forceinline__ void add_nulls_lane(int node) const { atomicAdd(&nullcnt[node], 1u); }
  // domain-0 bitmaps: rows == lanes, one 64-bit store per wavefront
  template <bool ACC>
  __device__ __forceinline__ void put_word0(int buf, uint64_t m) const {
    static_assert(!ACC, "the interpreter walks a wavefront's records in one piece");
    if (lane == 0 && wave_live) st_global<uint64_t, kWide>(this->buf(buf), lrow >> 6, m);
  }
  // child-domain bitmaps: one fire-and-forget atomic per bit on zeroed words (k_init).  The specialised kernels build
  // these words in LDS instead (spec_body.h SCtx::set_bit); this form is the fallback for rows beyond their LDS words.
  __device__ __forceinline__ void set_bit(int buf, int /*dom*/, uint32_t row) const {
    atomic_or_global(this->buf(buf), row >> 5, 1u << (row & 31));
  }
};

// --------------------------------------------------------------------------
// the interpreter: scalar program counter, one handler call per op
// --------------------------------------------------------------------------
// (round 6: the interpreter walks fast or careful like the specialised kernels -- walk.h `reject` -- instead of always careful)
// (round 6, last session: the program is read through the CONSTANT address space -- s_load_dwordx8 + x2 on a wave-uniform pc --
//  and the op that follows is requested before the current one runs.  Through the generic pointer of KParams the compiler fetched
//  every op with two VECTOR loads + v_readfirstlane behind s_waitcnt vmcnt(0): a vector-L1 round trip in front of every handler,
//  and in the emit walk -- vmcnt counts in order on this part -- a drain of every store the wavefront had in flight, once per op.)
// Handlers with the op's kind and its null-union bits as compile-time constants (the FAST walks only): h_fixed / h_string test
// `op.a`, `op.code` and `op.flags` a dozen times each, and with a run-time op every test is a scalar compare + select or branch --
// the generic kernels are bound by the CU's scalar unit (profiles/r06_s5_generic_scalar_program.txt: ~50 SALU instructions per
// op).  The dispatch picks the instance; inside it the tests fold.
template <int A, int NL, bool EMIT, class Ctx, class Src>      // NL: 0 = no null union, 1 = ["null", T], 2 = [T, "null"]
__device__ __forceinline__ void h_fixed_k(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  Op o = op;
  o.a = A;
  o.flags = (op.flags & ~(F_NULLABLE | F_NULL_FIRST)) | (NL == 1 ? (F_NULLABLE | F_NULL_FIRST) : NL == 2 ? F_NULLABLE : 0);
  h_fixed<EMIT, false>(c, src, L, o);      // (a further split on `dom == 0` in the emit walk: no gain, profiles/r06_s5_*)
}
template <int CODE, int NL, bool EMIT, class Ctx, class Src>
__device__ __forceinline__ void h_string_k(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  Op o = op;
  o.code = CODE;
  o.flags = (op.flags & ~(F_NULLABLE | F_NULL_FIRST)) | (NL == 1 ? (F_NULLABLE | F_NULL_FIRST) : NL == 2 ? F_NULLABLE : 0);
  h_string<EMIT, false>(c, src, L, o);
}
template <int A, bool EMIT, class Ctx, class Src>
__device__ __forceinline__ void h_fixed_nl(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const int nl = (op.flags & F_NULLABLE) ? ((op.flags & F_NULL_FIRST) ? 1 : 2) : 0;
  if (nl == 0) h_fixed_k<A, 0, EMIT>(c, src, L, op);
  else if (nl == 1) h_fixed_k<A, 1, EMIT>(c, src, L, op);
  else h_fixed_k<A, 2, EMIT>(c, src, L, op);
}
template <int CODE, bool EMIT, class Ctx, class Src>
__device__ __forceinline__ void h_string_nl(const Ctx& c, const Src& src, Lane& L, const Op& op) {
  const int nl = (op.flags & F_NULLABLE) ? ((op.flags & F_NULL_FIRST) ? 1 : 2) : 0;
  if (nl == 0) h_string_k<CODE, 0, EMIT>(c, src, L, op);
  else if (nl == 1) h_string_k<CODE, 1, EMIT>(c, src, L, op);
  else h_string_k<CODE, 2, EMIT>(c, src, L, op);
}

typedef const __attribute__((address_space(4))) Op* ProgPtr;
__device__ __forceinline__ Op ld_op(ProgPtr p) {      // (field by field: a struct copy out of another address space does not compile on the host pass)
  Op o;
  o.code = p->code; o.flags = p->flags; o.dom = p->dom; o.a = p->a; o.b = p->b; o.c = p->c;
  o.buf0 = p->buf0; o.buf1 = p->buf1; o.buf2 = p->buf2; o.node = p->node;
  return o;
}
template <bool EMIT, bool CAREFUL, class Ctx, class Src>
__device__ __forceinline__ void walk(const KParams& P, const Ctx& c, const Src& src, Lane& L) {
  const ProgPtr prog = reinterpret_cast<ProgPtr>(reinterpret_cast<uintptr_t>(P.prog));
  int pc = 0;
  Op nxt = ld_op(prog);
  for (;;) {
    pc = __builtin_amdgcn_readfirstlane(pc);
    const Op op = nxt;
    // (a use of every field HERE: without it the compiler sinks each field's load into the handler that reads it -- a dozen
    //  narrow s_loads, each waited for on the spot -- instead of one s_load_dwordx8 + x2 that is in flight while the previous op runs)
    asm volatile("" ::"s"(op.code), "s"(op.flags), "s"(op.dom), "s"(op.a), "s"(op.b), "s"(op.c), "s"(op.buf0), "s"(op.buf1), "s"(op.buf2), "s"(op.node));
    if (op.code == OP_END) return;                          // (the last op of a program: nothing is read behind it)
    int npc = op.code == OP_LIST_TAIL ? op.b : pc + 1;      // LIST_TAIL goes back to its LIST_NEXT
    nxt = ld_op(prog + npc);
    switch (op.code) {
      case OP_FIXED:
        if constexpr (CAREFUL) h_fixed<EMIT, true>(c, src, L, op);
        else {
          switch (op.a) {
            case FK_I32: h_fixed_nl<FK_I32, EMIT>(c, src, L, op); break;
            case FK_I64: h_fixed_nl<FK_I64, EMIT>(c, src, L, op); break;
            case FK_BOOL: h_fixed_nl<FK_BOOL, EMIT>(c, src, L, op); break;
            default: h_fixed<EMIT, false>(c, src, L, op); break;      // float / double
          }
        }
        break;
      case OP_STRING:
        if constexpr (CAREFUL) h_string<EMIT, true>(c, src, L, op);
        else h_string_nl<OP_STRING, EMIT>(c, src, L, op);
        break;
      case OP_ENUM:
        if constexpr (CAREFUL) h_string<EMIT, true>(c, src, L, op);
        else h_string_nl<OP_ENUM, EMIT>(c, src, L, op);
        break;
      case OP_REC_BEGIN: h_rec_begin<EMIT, CAREFUL>(c, src, L, op); break;
      case OP_REC_END: h_rec_end(L); break;
      case OP_UNION_BEGIN: h_union_begin<EMIT, CAREFUL>(c, src, L, op); break;
      case OP_VARIANT: h_variant(L, op); break;
      case OP_UNION_END: h_union_end(L); break;
      case OP_LIST_BEGIN: h_list_begin<EMIT, CAREFUL>(c, src, L, op); break;
      case OP_LIST_NEXT:
        if (!h_list_next<CAREFUL, (EMIT && !CAREFUL)>(c, src, L, op)) { npc = op.b; nxt = ld_op(prog + npc); }      // no lane has an item left: to LIST_END
        break;
      case OP_LIST_TAIL: h_list_tail(c, L, op); break;
      case OP_LIST_END: h_list_end<EMIT>(c, L, op); break;
      case OP_BIN: h_bin<EMIT, CAREFUL>(c, src, L, op); break;
      default: return;
    }
    pc = npc;
  }
}

// --------------------------------------------------------------------------
// LDS carving shared by k_size / k_emit
// --------------------------------------------------------------------------
struct Smem {
  uint32_t* cnt;      // [KL][T]   per-lane counters (all K of them unless the schema is wide)
  uint32_t* rem;      // [list_depth][T]
  uint32_t* wtot;     // [K][NW]
  uint32_t* gb;       // [K]   chunk-relative workgroup base
  uint32_t* wv;       // [K]   wide schemas only (KL < K): ICtx::wv
  uint32_t* nullcnt;  // [nnodes]
  uint32_t* misc;     // [4]: 0 = lowest erroring tid
  uint64_t* bufs;     // [nbuf]
  uint8_t* win;       // input window (16-byte aligned, +16 bytes of slack)
};

__host__ __device__ inline uint32_t lds_fixed_words(int K, int KL, int T, int list_depth, int nnodes, int nbuf) {
  const uint32_t k1 = (uint32_t)(K > 0 ? K : 1), kl1 = (uint32_t)(KL > 0 ? KL : 1);
  const uint32_t k4 = (k1 + 3) & ~3u;
  return kl1 * (uint32_t)T + (uint32_t)(list_depth > 0 ? list_depth : 1) * (uint32_t)T + ((k1 * (uint32_t)(T / 64) + 3) & ~3u) + k4 + (KL < K ? k4 : 0u) +
         (uint32_t)((nnodes + 3) & ~3) + 4 + 2 * (uint32_t)((nbuf + 1) & ~1);
}

template <int T>
__device__ __forceinline__ Smem carve(const KParams& P, uint8_t* smem) {
  Smem s;
  const uint32_t k1 = (uint32_t)(P.K > 0 ? P.K : 1), kl1 = (uint32_t)(P.KL > 0 ? P.KL : 1);
  const uint32_t k4 = (k1 + 3) & ~3u;
  uint32_t* p = reinterpret_cast<uint32_t*>(smem);
  s.cnt = p; p += kl1 * T;
  s.rem = p; p += (P.list_depth > 0 ? P.list_depth : 1) * T;
  s.wtot = p; p += (k1 * (T / 64) + 3) & ~3u;
  s.gb = p; p += k4;
  s.wv = p; p += P.KL < P.K ? k4 : 0u;
  s.nullcnt = p; p += ((P.nnodes + 3) & ~3);
  s.misc = p; p += 4;
  s.bufs = reinterpret_cast<uint64_t*>(p); p += 2 * ((P.nbuf + 1) & ~1);
  s.win = reinterpret_cast<uint8_t*>(p);          // all pieces above are multiples of 16 bytes
  return s;
}

// Host mirror of carve(): LDS bytes in front of the window.
extern "C" uint32_t rh_lds_fixed_bytes(int K, int KL, int tile, int list_depth, int nnodes, int nbuf) {
  return lds_fixed_words(K, KL, tile, list_depth, nnodes, nbuf) * 4;
}

// fits: the tile's window is staged in LDS; cursors become LDS byte addresses (walk.h LdsAbsSrc).  A tile past the window is
// walked carefully, straight from global memory (this form has no ranges: spec_body.h ranged_tile is the specialised kernels').
template <bool EMIT, bool CAREFUL, int T>
__device__ __forceinline__ void run_walk(const KParams& P, const ICtx<T>& c, const Smem& s, Lane& L, bool fits, uint64_t wb16) {
  if (fits) {
    const uint32_t wa = (uint32_t)(uintptr_t)(RH_LDS uint8_t*)s.win;
    L.cur += wa; L.end += wa;
    LdsAbsSrc src;
    walk<EMIT, CAREFUL>(P, c, src, L);
  } else {
    GlobalSrc src{P.data + wb16, P.data_len - wb16};
    walk<EMIT, true>(P, c, src, L);
  }
}

template <int T>
__device__ __forceinline__ ICtx<T> make_ctx(const KParams& P, const Smem& s, const Geo& g, uint32_t tid) {
  ICtx<T> c;
  c.cnt = s.cnt; c.rem = s.rem; c.nullcnt = s.nullcnt; c.bufs = s.bufs; c.gb = s.gb; c.wv = s.wv;
  c.sym_off = P.sym_off; c.sym_data = P.sym_data;
  c.lrow = g.lrow0 + tid; c.tid = tid; c.lane = tid & 63; c.wave_live = ((tid >> 6) * 64) < g.nrec;
  return c;
}

// --------------------------------------------------------------------------
// k_size.  T = records (= threads) per tile: 256, or 64 for a wide schema (program.h kWideTile: one wavefront per tile, so that
// the byte counters of domain-0 columns are wave counters with no per-lane storage -- ICtx::wave_total / wave_offset)
// --------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void k_size_body(const KParams& P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NW = T / 64;
  const Smem s = carve<T>(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = geometry<T>(P, blockIdx.x);
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (fits) stage_window<T>(P, s.win, wb16, we, tid);
  for (int k = 0; k < P.KL; k++) s.cnt[k * T + tid] = 0;
  if (tid == 0) { s.misc[0] = 0xFFFFFFFFu; s.misc[2] = 0; }
  __syncthreads();

  Lane L;
  lane_init(L, P, g, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;   // window beyond 32-bit cursors
  const ICtx<T> c = make_ctx<T>(P, s, g, tid);
  uint32_t tflag = fits ? 0u : (uint32_t)(TF_OVER_WINDOW | TF_CAREFUL);
  if (fits) {
    // the fast walk; a wavefront with a record outside the fast wire forms (or malformed) walks again, carefully
    run_walk<false, false>(P, c, s, L, true, wb16);
    L.redo = L.redo || L.cur > L.end;
    if (__any(L.redo)) {
      tflag |= (uint32_t)(TF_CAREFUL | TF_REWALK_ONE);
      for (int k = 0; k < P.KL; k++) s.cnt[k * T + tid] = 0;
      for (int d = 0; d < P.list_depth; d++) s.rem[d * T + tid] = 0;
      lane_init(L, P, g, wb16, tid);
      run_walk<false, true>(P, c, s, L, true, wb16);
    }
  } else {
    run_walk<false, true>(P, c, s, L, false, wb16);
  }
  if (lane == 0 && tflag) { atomicOr(&s.misc[2], tflag & ~(uint32_t)TF_REWALK_ONE); if (tflag & (uint32_t)TF_REWALK_ONE) atomicAdd(&s.misc[2], (uint32_t)TF_REWALK_ONE); }

  // every record's per-lane counters -> HBM ([tile][counter][record]: coalesced), so that k_emit does not size again
  for (int k = 0; k < P.KL; k++) {
    const uint32_t cv = s.cnt[k * T + tid];
    P.lanecnt[((size_t)blockIdx.x * P.KL + k) * T + tid] = cv;
    const uint32_t v = wave_sum(cv);
    if (lane == 0) s.wtot[k * NW + wave] = v;
  }
  report_errors(P, s.misc, L, g, tid, blockIdx.x);   // contains the barrier that publishes wtot, misc[2] (and the wave counters' sums in wv)
  if (tid == 0) P.tileflag[blockIdx.x] = s.misc[2];
  for (int k = tid; k < P.K; k += T) {
    uint32_t t = 0;
    if (k < P.KL) { for (int w = 0; w < NW; w++) t += s.wtot[k * NW + w]; }
    else t = s.wv[k];                                 // (a wide schema's tile is one wavefront: its sum is the tile's)
    P.blocksum[(size_t)k * P.nblocks + blockIdx.x] = t;
  }
}
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_size(KParams P) { k_size_body<kBlock>(P); }
extern "C" __global__ void __launch_bounds__(kWideTile) rh_k_size_w(KParams P) { k_size_body<kWideTile>(P); }

// --------------------------------------------------------------------------
// k_scan: one workgroup per (counter, chunk)
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_scan(KParams P);

// --------------------------------------------------------------------------
// k_scan + k_layout in ONE launch (the single-submission path): every workgroup scans its (counter, chunk) segment
// like rh_k_scan; the workgroup that finishes LAST (ticket in control word [1] high half, zeroed with the control
// words) lays the arena out from the chunk totals all of them left.  One launch and one inter-kernel gap less per
// call -- what a 1M-record call is made of (profiles/r03e_*).  Cross-workgroup visibility of the totals: release
// fence + ticket by the writer, acquire fence by the last workgroup (agent scope; the L2s hold next to nothing dirty
// here, so the fences are cheap -- unlike in a tile look-back).
// --------------------------------------------------------------------------
// kScanPer consecutive tiles per thread and round (serial prefix inside the thread, DPP scan of the thread sums across
// the wave, four wave totals through LDS).  Long segments take 8 per thread: a 4883-tile chunk (1.25M records, what
// ONE rank of the 8-GPU configuration scans in a single workgroup per counter) is 3 rounds of two barriers instead of
// 20 (rh_k_scan_layout 17 -> 9 us there, 22 -> 19 us at 10M records / 8 chunks); short ones keep one tile per thread
// (a 489-tile segment of a 1M-record call was 2 us slower with 8).
template <uint32_t kScanPer>
__device__ __forceinline__ void scan_segment_n(const KParams& P, uint32_t* wt) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t kk = blockIdx.x % (uint32_t)P.K, ch = blockIdx.x / (uint32_t)P.K;
  const uint32_t b0 = ch * P.bpc;
  const uint32_t b1 = ch == P.k - 1 ? P.nblocks : (ch + 1) * P.bpc;
  const uint32_t* in = P.blocksum + (size_t)kk * P.nblocks;
  uint32_t* out = P.blockbase + (size_t)kk * P.nblocks;
  uint64_t carry = 0;
  for (uint32_t base = b0; base < b1; base += kBlock * kScanPer) {
    const uint32_t i0 = base + tid * kScanPer;
    uint32_t v[kScanPer];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; j++) {
      v[j] = i0 + j < b1 ? in[i0 + j] : 0;
      sum += v[j];
    }
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) wt[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wt[w];
    const uint32_t round = wt[0] + wt[1] + wt[2] + wt[3];
    uint32_t run = (uint32_t)(carry + woff + (incl - sum));
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; j++) {
      if (i0 + j < b1) out[i0 + j] = run;
      run += v[j];
    }
    carry += round;
    __syncthreads();
  }
  if (tid == 0) P.totals[(size_t)kk * P.k + ch] = carry;
}

__device__ __forceinline__ void scan_segment(const KParams& P, uint32_t* wt) {
  if (P.bpc > 1024u) scan_segment_n<8>(P, wt);       // (uniform for the launch)
  else scan_segment_n<1>(P, wt);
}

extern "C" __global__ void __launch_bounds__(kBlock) rh_k_scan(KParams P) {
  __shared__ uint32_t wt[4];
  scan_segment(P, wt);
}

// --------------------------------------------------------------------------
// k_layout: one workgroup.  Entry e = chunk * nbuf + buf of the [chunk][buf] tables gets its arena slot: an exclusive
// scan of the slot sizes in table order (each thread owns a contiguous run of entries, the 256 run sums are scanned
// in LDS).  Replaces the host round trip (totals to the host, layout there, tables back) of round 1.
// --------------------------------------------------------------------------
__device__ __forceinline__ uint64_t layout_entry(const LParams& L, uint32_t e, uint32_t& flag) {
  const uint32_t c = e / (uint32_t)L.nbuf, b = e - c * (uint32_t)L.nbuf;
  const BufDesc d = L.desc[b];
  const uint64_t rows0 = c == L.k - 1 ? L.rows_last : L.sz;
  const uint64_t rows = d.dom == 0 ? rows0 : L.totals[(size_t)(d.dom - 1) * L.k + c];
  const uint64_t tot = d.kind == BK_DATA ? L.totals[(size_t)d.counter * L.k + c] : 0;
  if (tot > 0x7FFFFFFFull || (d.dom != 0 && rows > 0x7FFFFFFFull)) flag |= LF_OFFSET32;
  if (L.narrow && d.dom != 0 && rows >= L.narrow_rows) flag |= LF_NEED_WIDE;
  return buf_bytes(d.kind, rows, tot, nullptr, (uint32_t)d.counter);
}

// the layout of one call, by the 256 threads of ONE workgroup (rh_k_layout, or the last workgroup of rh_k_scan_layout)
__device__ __forceinline__ void layout_body(const LParams& L, uint64_t* run_sum, uint32_t* flags_p) {
  uint32_t& flags = *flags_p;
  const uint32_t tid = threadIdx.x;
  const uint32_t E = L.k * (uint32_t)L.nbuf;
  const uint32_t per = (E + kBlock - 1) / kBlock;
  const uint32_t e0 = tid * per < E ? tid * per : E, e1 = e0 + per < E ? e0 + per : E;
  if (tid == 0) flags = 0;
  __syncthreads();
  uint32_t flag = 0;
  uint64_t mine = 0;
  for (uint32_t e = e0; e < e1; e++) mine += buf_slot_bytes(layout_entry(L, e, flag));
  // exclusive scan of the 256 run sums: 64-bit shuffle scan inside each wave, the four wave totals through LDS
  // (a serial scan by thread 0 was ~2 us of the 7 us this kernel takes on its own -- a 1M-record flat-schema call)
  const uint32_t lane = tid & 63, wave = tid >> 6;
  uint64_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t up = __shfl_up(incl, d, 64);
    if ((int)lane >= d) incl += up;
  }
  if (lane == 63) run_sum[wave] = incl;
  if (flag) atomicOr(&flags, flag);
  __syncthreads();
  uint64_t wbase = 0;
  for (uint32_t w = 0; w < wave; w++) wbase += run_sum[w];
  if (tid == kBlock - 1) {
    const uint64_t acc = wbase + incl;
    const uint64_t used = acc < kBufAlign ? kBufAlign : acc;
    uint32_t f = flags | (reinterpret_cast<uint32_t*>(L.ctrl)[2] & (uint32_t)LF_NEED_RANGED);      // (the size kernel's verdict stays)
    if (used > L.capacity) f |= LF_CAPACITY;
    L.ctrl[2] = used;
    reinterpret_cast<uint32_t*>(L.ctrl)[2] = f;
    flags = f;
  }
  __syncthreads();
  if (flags) return;       // nothing may be written through these tables: leave them alone
  uint64_t off = wbase + incl - mine;
  uint32_t dummy = 0;
  for (uint32_t e = e0; e < e1; e++) {
    const uint64_t sz = layout_entry(L, e, dummy);
    L.bufptr[e] = L.arena + off;
    L.bufsize[e] = sz;
    // offsets[0] = 0 of every offsets buffer (what k_init does on the exact path; k_init then only has the
    // atomically-built child-domain bitmaps left to zero, and is not launched at all for schemas without any)
    if (L.desc[e % (uint32_t)L.nbuf].kind == BK_OFFSETS) *reinterpret_cast<uint32_t*>(L.arena + off) = 0;
    off += buf_slot_bytes(sz);
  }
}

extern "C" __global__ void __launch_bounds__(kBlock) rh_k_layout(LParams L) {
  __shared__ uint64_t run_sum[kBlock];
  __shared__ uint32_t flags;
  layout_body(L, run_sum, &flags);
}

extern "C" __global__ void __launch_bounds__(kBlock) rh_k_scan_layout(KParams P, LParams L) {
  __shared__ uint64_t run_sum[kBlock];
  __shared__ uint32_t wt[4];
  __shared__ uint32_t flags;
  __shared__ uint32_t last;
  scan_segment(P, wt);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this workgroup's total before its ticket
    uint32_t* ticket = reinterpret_cast<uint32_t*>(L.ctrl) + 3;
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == gridDim.x - 1 ? 1u : 0u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // everybody's totals before the layout reads them
  }
  __syncthreads();
  if (!last) return;
  layout_body(L, run_sum, &flags);
}

// --------------------------------------------------------------------------
// k_publish: the last launch of a call.  Hands the call's control block (first_bad, layout flag, arena bytes, null
// counts, chunk totals -- a few hundred bytes to ~2 KB) to the host by storing it straight into pinned host memory,
// and leaves the device copy ZEROED for the next call that leases it -- one ~2 us launch in place of a D2H blit kernel
// (4.5 us) plus the fill kernel (1.7 us) that re-zeroed the block, which were 4-15 % of a 1M-record call
// (profiles/r03s_timeline_*.txt).
// --------------------------------------------------------------------------
// The null counts leave as ONE word per (node, chunk): the kNullSlots addresses the workgroups added into are summed here.
// The LAST store is a token at host[flag_word] (system scope, behind a system-scope fence of every thread's stores and a
// barrier): the host spins on that word in its pinned block instead of waiting for a stream event -- an event record
// between two calls cost the GPU 5.7 us of idle time per call (profiles/r03aj_timeline.txt), and the spin sees the
// token ~3 us sooner than hipStreamSynchronize returns (tools/synclat.hip).
// ABI 7: the size pass's per-tile flags (program.h TileFlag) are summed here into four words of the head -- what
// rh_engine_counters reports as careful / over-window / sub-tiled tiles and re-walked wavefronts (nflags = 0: no size pass ran).
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_publish(uint32_t* ctrl, uint32_t* host, uint32_t head_words,
                                                                 uint32_t null_entries, uint32_t flag_word, uint32_t token, uint32_t nslots,
                                                                 const uint32_t* tileflag, uint32_t nflags, uint32_t stat_word) {
  __shared__ uint32_t stat[4];
  if (threadIdx.x < 4) stat[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < head_words; i += kBlock) {      // control words + chunk totals, as they are
    host[i] = ctrl[i];
    ctrl[i] = 0;
  }
  if (nflags) {
    uint32_t car = 0, over = 0, sub = 0, rew = 0;
    for (uint32_t i = threadIdx.x; i < nflags; i += kBlock) {
      const uint32_t f = tileflag[i];
      car += (f >> 1) & 1u; over += (f >> 2) & 1u; sub += (f >> 3) & 1u; rew += (f >> 8) & 0xFFu;
    }
    car = wave_sum(car); over = wave_sum(over); sub = wave_sum(sub); rew = wave_sum(rew);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stat[0], car); atomicAdd(&stat[1], over); atomicAdd(&stat[2], rew); atomicAdd(&stat[3], sub); }
    __syncthreads();
    if (threadIdx.x < 4) host[stat_word + threadIdx.x] = stat[threadIdx.x];
  }
  uint32_t* slots = ctrl + head_words;                               // [null_entries][nslots] (program.h null_slots_for)
  for (uint32_t e = threadIdx.x; e < null_entries; e += kBlock) {
    uint32_t sum = 0;
    if (nslots == (uint32_t)kNullSlots) {
      const v4w* p = reinterpret_cast<const v4w*>(slots + (size_t)e * kNullSlots);
#pragma unroll
      for (int q = 0; q < kNullSlots / 4; q++) { const v4w x = p[q]; sum += x.x + x.y + x.z + x.w; }
    } else {
      for (uint32_t q = 0; q < nslots; q++) sum += slots[(size_t)e * nslots + q];
    }
    host[head_words + e] = sum;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < null_entries * nslots; i += kBlock) slots[i] = 0;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host + flag_word, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// true when the call must not emit: a malformed record was found by the size pass, or the layout kernel said no
__device__ __forceinline__ bool call_aborted(const unsigned long long* ctrl) {
  return ctrl[0] != 0 || reinterpret_cast<const uint32_t*>(ctrl)[2] != 0;
}

// --------------------------------------------------------------------------
// k_init: offsets[0] = 0 for every offsets buffer, zero the atomically-built bitmaps.
// grid = k * nbuf workgroups ([chunk][buf] tables)
// --------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_init(void* const* bufptr, const uint64_t* bufsize,
                                                              const BufDesc* desc, uint32_t nbuf, uint32_t k,
                                                              const unsigned long long* ctrl) {
  if (call_aborted(ctrl)) return;
  const uint32_t id = blockIdx.x % nbuf;
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
template <int T>
__device__ __forceinline__ void k_emit_body(const KParams& P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NW = T / 64;
  // Uniform for the workgroup: first_bad is only honoured when a size pass ran (K > 0) -- it is final before this kernel
  // starts then; without a size pass this kernel is the one that writes it, and waves of one workgroup reading it at
  // different times could disagree about returning ahead of the barriers below.  The layout flag is always final here.
  if ((P.K > 0 && P.first_bad[0] != 0) || reinterpret_cast<const uint32_t*>(P.first_bad)[2] != 0) return;
  const Smem s = carve<T>(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = geometry<T>(P, blockIdx.x);
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (fits) stage_window<T>(P, s.win, wb16, we, tid);
  for (int k = 0; k < P.KL; k++) s.cnt[k * T + tid] = 0;
  for (int i = tid; i < P.nnodes; i += T) s.nullcnt[i] = 0;
  for (int i = tid; i < P.nbuf; i += T)
    s.bufs[i] = reinterpret_cast<uint64_t>(P.bufptr[(size_t)g.chunk * P.nbuf + i]);
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  __syncthreads();

  Lane L;
  const ICtx<T> c = make_ctx<T>(P, s, g, tid);

  // the size pass's verdict on this tile (no size pass -- K == 0 -- or RUHVRO_HIP_NO_TRUST: every tile carefully)
  const bool careful = !fits || P.all_careful != 0 || (P.K > 0 && (P.tileflag[blockIdx.x] & (uint32_t)TF_CAREFUL) != 0);
  if (P.K > 0) {
    // the per-lane counters as k_size left them (round 6: 4 bytes per counter and record each way instead of a second size
    // walk -- the interpreter's walk costs ~100 instructions per op), then the in-workgroup scan
    // (the wave counters of a wide schema need neither: ICtx::wave_offset scans them on the spot in the emit walk)
    if (P.KL > 0) {
      for (int k = 0; k < P.KL; k++) {
        const uint32_t v = P.lanecnt[((size_t)blockIdx.x * P.KL + k) * T + tid];
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s.wtot[k * NW + wave] = incl;
        s.cnt[k * T + tid] = incl - v;
      }
      __syncthreads();
      if (NW > 1) {
        for (int k = 0; k < P.KL; k++) {
          uint32_t base = 0;
          for (uint32_t w = 0; w < wave; w++) base += s.wtot[k * NW + w];
          s.cnt[k * T + tid] += base;             // workgroup-local exclusive prefix
        }
      }
    }
    for (int k = tid; k < P.K; k += T) {
      const uint32_t b = P.blockbase[(size_t)k * P.nblocks + blockIdx.x];
      s.gb[k] = b;
      if (k >= P.KL) s.wv[k] = b;                 // the running base of a wave counter starts at the tile's base
    }
    __syncthreads();
  }

  lane_init(L, P, g, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;
  if (careful) {
    run_walk<true, true>(P, c, s, L, fits, wb16);
  } else {
    run_walk<true, false>(P, c, s, L, true, wb16);     // (trusted: the size pass met no anomaly in this tile)
    if (L.redo) L.err = E_INTERNAL;
  }

  report_errors(P, s.misc, L, g, tid, blockIdx.x);   // barrier inside: nullcnt + staging complete
  for (int i = tid; i < P.nnodes; i += T) {
    const uint32_t v = s.nullcnt[i];
    if (v) atomicAdd(&P.nullcount[((size_t)i * P.k + g.chunk) * P.null_slots + (blockIdx.x & (P.null_slots - 1))], v);
  }
}
extern "C" __global__ void __launch_bounds__(kBlock) rh_k_emit(KParams P) { k_emit_body<kBlock>(P); }
extern "C" __global__ void __launch_bounds__(kWideTile) rh_k_emit_w(KParams P) { k_emit_body<kWideTile>(P); }

}  // namespace rh

// --------------------------------------------------------------------------
// launchers (called from engine.cpp; plain C linkage, HIP types stay in here)
// --------------------------------------------------------------------------
extern "C" int rh_launch_size(const rh::KParams* P, uint32_t lds_bytes, void* stream, void* start, void* stop) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  if (P->tile == (uint32_t)rh::kWideTile)
    hipExtLaunchKernelGGL(rh::rh_k_size_w, dim3(P->nblocks), dim3(rh::kWideTile), lds_bytes, (hipStream_t)stream, (hipEvent_t)start, (hipEvent_t)stop, 0, *P);
  else
  hipExtLaunchKernelGGL(rh::rh_k_size, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, (hipEvent_t)start,
                        (hipEvent_t)stop, 0, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_scan(const rh::KParams* P, void* stream, void* start, void* stop) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipExtLaunchKernelGGL(rh::rh_k_scan, dim3((uint32_t)P->K * P->k), dim3(rh::kBlock), 0, (hipStream_t)stream, (hipEvent_t)start,
                        (hipEvent_t)stop, 0, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_scan_layout(const rh::KParams* P, const rh::LParams* L, void* stream, void* start, void* stop) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipExtLaunchKernelGGL(rh::rh_k_scan_layout, dim3((uint32_t)P->K * P->k), dim3(rh::kBlock), 0, (hipStream_t)stream, (hipEvent_t)start,
                        (hipEvent_t)stop, 0, *P, *L);
  return (int)hipGetLastError();
}

extern "C" int rh_launch_init(void* const* bufptr, const uint64_t* bufsize, const rh::BufDesc* desc, uint32_t nbuf,
                              uint32_t k, const unsigned long long* ctrl, void* stream) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipLaunchKernelGGL(rh::rh_k_init, dim3(nbuf * k), dim3(rh::kBlock), 0, (hipStream_t)stream, bufptr, bufsize, desc,
                     nbuf, k, ctrl);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_publish(void* ctrl, void* host, uint32_t head_words, uint32_t null_entries, uint32_t flag_word, uint32_t token,
                                 uint32_t nslots, const uint32_t* tileflag, uint32_t nflags, uint32_t stat_word, void* stream) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipLaunchKernelGGL(rh::rh_k_publish, dim3(1), dim3(rh::kBlock), 0, (hipStream_t)stream, (uint32_t*)ctrl, (uint32_t*)host, head_words,
                     null_entries, flag_word, token, nslots, tileflag, nflags, stat_word);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_layout(const rh::LParams* L, void* stream) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipLaunchKernelGGL(rh::rh_k_layout, dim3(1), dim3(rh::kBlock), 0, (hipStream_t)stream, *L);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_emit(const rh::KParams* P, uint32_t lds_bytes, void* stream, void* start, void* stop) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  if (P->tile == (uint32_t)rh::kWideTile)
    hipExtLaunchKernelGGL(rh::rh_k_emit_w, dim3(P->nblocks), dim3(rh::kWideTile), lds_bytes, (hipStream_t)stream, (hipEvent_t)start, (hipEvent_t)stop, 0, *P);
  else
  hipExtLaunchKernelGGL(rh::rh_k_emit, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, (hipEvent_t)start,
                        (hipEvent_t)stop, 0, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_set_max_lds(uint32_t bytes) {
  for (const void* f : {reinterpret_cast<const void*>(rh::rh_k_size), reinterpret_cast<const void*>(rh::rh_k_emit),
                        reinterpret_cast<const void*>(rh::rh_k_size_w), reinterpret_cast<const void*>(rh::rh_k_emit_w)}) {
    const int e = (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e) return e;
  }
  return 0;
}
