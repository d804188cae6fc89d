// Per-schema specialised kernels of the direct-decode path (gfx950).
//
// specialize.cpp turns a compiled schema into a tiny translation unit: a struct
// `Spec` with the schema's constants and a `walk<EMIT>` member that calls the
// field handlers of walk.h in schema order with constexpr Ops, plus two kernel
// stubs that instantiate the bodies below.  With every Op a compile-time
// constant the handlers fold to straight-line code per field: no op fetch, no
// dispatch, and the per-lane counters (child rows / string bytes) live in
// registers instead of LDS, which is what lets several workgroups share a CU.
//
// Same algorithm, same launch sequence and same KParams as the generic kernels
// in kernels.hip (k_size -> k_scan -> k_init -> k_emit); only k_size / k_emit
// are replaced.
#pragma once
#include "kernel_common.h"

namespace rh {

// compile-time loop: every array index below is a constant, so the arrays scalarise into registers
template <int I>
struct IC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

// Phase timing (RUHVRO_HIP_PROFILE=1): lane 0 of every wave adds the shader-clock cycles between
// marks to P.prof[slot]; the host prints per-wave means.  Compiled out otherwise.
#ifdef RH_PROFILE
#define RH_MARK(slot)                                   \
  do {                                                  \
    const unsigned long long _t = clock64();            \
    _pd[slot] = _t - _tprev;                            \
    _tprev = _t;                                        \
  } while (0)
#define RH_MARK_INIT                                    \
  unsigned long long _pd[20];                           \
  for (int _i = 0; _i < 20; _i++) _pd[_i] = 0;          \
  unsigned long long _tprev = clock64()
// one fire-and-forget atomic per slot per wave, spread over 64 rows to keep L2 contention out of the picture
#define RH_MARK_FLUSH                                                                                     \
  do {                                                                                                    \
    if ((threadIdx.x & 63) == 0 && P.prof) {                                                              \
      unsigned long long* _row = P.prof + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63) * 32;      \
      for (int _i = 0; _i < 20; _i++)                                                                     \
        if (_pd[_i]) atomicAdd(&_row[_i], _pd[_i]);                                                       \
    }                                                                                                     \
  } while (0)
#else
#define RH_MARK(slot) do {} while (0)
#define RH_MARK_INIT do {} while (0)
#define RH_MARK_FLUSH do {} while (0)
#endif

constexpr int kBmWords = 64;      // LDS words per child-domain bitmap and tile (2048 rows before the global fallback)
constexpr int kDirectLanes = 24;  // a tile past the window whose ranges would hold fewer records than this on average is walked from global memory in one piece (ranged_tile)
constexpr int kScanSets = 4;      // item_scan: candidate starts per step = 64 x this (a span of 256 bytes)
constexpr uint32_t kRangeSlack = 64; // bytes staged behind a range's last record (ranged_tile)
constexpr int kDenseCap = 128;    // item positions per wavefront and round of the item-dense list handling (dense_list)

template <class S>
struct SCtx {
  static constexpr int K1 = S::K > 0 ? S::K : 1;
  static constexpr int KP = (K1 + 3) & ~3;           // words per wavefront row of wtot
  // Wide schemas (schema.cpp: more than kWideCounters counters; TILE = 64): only the first KL counters -- row domains and the byte
  // columns inside arrays / maps -- live per lane; the byte columns of domain 0 behind them are WAVE counters (walk.h F_WAVE_CTR)
  static constexpr int KL = S::KL;
  static constexpr int KL1 = KL > 0 ? KL : 1;
  static constexpr int KLP = (KL1 + 3) & ~3;
  static constexpr bool kWaveCtr = S::KL < S::K;
  static_assert(!kWaveCtr || S::TILE == 64, "wave counters need tiles of one wavefront");
  uint32_t* wv;                                       // LDS [K] (wide): size walk = this walk's wavefront sum per wave counter;
                                                      //                 emit walk = the wavefront's running chunk-relative base
  __device__ __forceinline__ void wave_total(int id, uint32_t len) const {
    const uint32_t t = wave_sum(len);
    if (lane == 0) wv[id] = t;                        // (every wave counter is met once per walk: a plain store; committed by the caller)
  }
  __device__ __forceinline__ uint32_t wave_offset(int id, uint32_t len) const {
    const uint32_t incl = wave_incl_scan(len, lane);
    const uint32_t base = wv[id];
    if (lane == 63) wv[id] = base + incl;             // (DS operations of a wavefront execute in order: every lane has read `base`)
    return base + incl - len;
  }
  static constexpr bool kWide = false;   // 32-bit byte offsets into each buffer (host guards: buffers < 4 GiB per chunk)
  static constexpr bool kSkip = false;
  static constexpr bool kDense = S::NDENSE > 0;
  uint32_t* dtab;                                     // LDS [kDenseCap]: this wavefront's item-position table (dense_list)
  const uint32_t* wtot_w;                             // LDS: this wavefront's row of wtot: wtot_w[k] = its total of counter k
  mutable uint32_t cnt[KL1];                          // per-lane counters (registers)
  mutable uint32_t rem[S::DEPTH > 0 ? S::DEPTH : 1];  // items left in the current block, per list depth
  // this chunk's row of the buffer-address table.  Read-only for the whole launch, so it is addressed through the
  // constant address space: every use is a scalar load the compiler may re-issue instead of keeping (and spilling)
  // 2 x NBUF SGPRs -- with the addresses held in registers a third of the emit kernel's VALU instructions were
  // v_readlane reloads of spilled SGPRs.
  const __attribute__((address_space(4))) uint64_t* bufp;
  uint32_t gb[KL1];                                   // chunk-relative base of this workgroup per (per-lane) counter
  uint32_t* nullcnt;                                  // LDS [NNODES]: null rows of child domains (one atomic per null row)
  uint32_t* nullw_w;                                  // LDS [NNODES][NW] + this wave: null rows of domain-0 fields, per wavefront
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  uint32_t lrow, lane;
  bool wave_live;

  __device__ __forceinline__ uint32_t& counter(int id) const { return cnt[id]; }
  __device__ __forceinline__ uint32_t& remaining(int d) const { return rem[d]; }
  __device__ __forceinline__ void* buf(int id) const { return reinterpret_cast<void*>(bufp[id]); }
  __device__ __forceinline__ uint32_t gbase(int id) const { return gb[id]; }
  static constexpr bool kEnumImm = true;              // enum symbols as immediates where the schema allows (Spec::enum_sym)
  static __device__ __forceinline__ bool enum_sym(int b, uint32_t v, uint32_t& len, uint64_t& bits) { return S::enum_sym(b, v, len, bits); }
  // no per-node accumulators (NNODES wave-uniform registers that spill): one LDS add per field and wave
  // (k_emit 0.950 -> 0.937 ms, profiles/r02a_variants_ab.txt)
  // A domain-0 field is visited exactly once per wavefront and tile, so its null count of the wave is a plain store into
  // the wave's own word (summed per tile at the flush), not an atomic add: the compiler expands a wave-uniform LDS atomic
  // into ~19 instructions (its atomic optimiser: mbcnt / bcnt / mul around a single-lane ds_add) per nullable field.
  // (ACC: a wavefront that walks its records in several ranges -- ranged_tile -- adds to the word, zeroed at the tile's start)
  template <bool ACC>
  __device__ __forceinline__ void add_nulls_wave(int node, uint32_t n) const {
    if (lane == 0) {
      if constexpr (ACC) nullw_w[node * (S::TILE / 64)] += n;
      else nullw_w[node * (S::TILE / 64)] = n;
    }
  }
  __device__ __forceinline__ void add_nulls_lane(int node) const { atomicAdd(&nullcnt[node], 1u); }
  // Bits of child-domain bitmaps (validity / boolean values of list items, map values ...).  A tile's rows of a child
  // domain are one contiguous run [gb, gb + rows): its bitmap words are assembled in LDS (ds_or, kBmWords words per
  // bitmap, bit 0 = row gb & ~31) and flushed once per tile with one global atomicOr per non-zero WORD (spec_emit) --
  // instead of one global atomic per BIT, which serialises at the L2 when 32 lanes hit one dword (tools/storecost.hip:
  // ~1800 cycles per wave instruction; measured on a schema with nullable list items and map values, 2M records:
  // k_emit 0.354 -> 0.113 ms, profiles/r02f_child_bitmap_ab.jsonl).  Rows beyond the LDS words (a tile with > ~2000 child rows) take the global form.
  // domain-0 bitmaps: rows == lanes, one ballot and one 64-bit store per wavefront and field.  (Collecting the ~15 words
  // of a wave in one VGPR with v_writelane and storing them once was measured: no gain, profiles/r02g_variants_ab.txt.)
  template <bool ACC>
  __device__ __forceinline__ void put_word0(int buf, uint64_t m) const {
    // the tile's domain-0 bitmap words are collected in LDS and stored by ONE instruction per workgroup (spec_emit tail)
    // instead of one single-lane store per field and wave: a store costs the vector-memory path ~16 cycles even with one
    // active lane (tools/storecost.hip), 11-14 such words per wave on the benchmark schema (k_emit -4.2 %,
    // profiles/r03y_variants_ab.txt; the per-WAVE collection of round 2 -- v_writelane, one store per wave -- had not paid)
    if (lane == 0) {
      uint64_t& w = bmw0[S::bm0slot(buf) * (S::TILE / 64) + (lrow & (S::TILE - 1)) / 64];
      if constexpr (ACC) w |= m;      // (ranged_tile: the bits of this range's records onto the zeroed word)
      else w = m;
    }
  }
  uint64_t* bmw0;                                     // LDS [NB0][NW]: this tile's domain-0 bitmap words
  uint32_t* bm;                                       // LDS [NBM][kBmWords]
  __device__ __forceinline__ void set_bit(int buf, int dom, uint32_t row) const {
    if constexpr (S::NBM > 0) {
      const uint32_t bit = row - (gb[dom - 1] & ~31u);
      if (bit < (uint32_t)(kBmWords * 32)) {
        atomicOr(&bm[S::bmslot(buf) * kBmWords + (bit >> 5)], 1u << (bit & 31));
        return;
      }
    }
    atomic_or_global(this->buf(buf), row >> 5, 1u << (row & 31));
  }
};

// Per-record counters handed from the size pass to the emit pass: RECORD-major, two 16-bit counters per dword, so a
// lane moves its K counters with ceil(K/2) dwords in 16 / 8 / 4-byte pieces (two instructions for the benchmark
// schema's 12 counters) instead of K two-byte accesses at a [counter][record] layout.
template <int NDW, int I = 0>
__device__ __forceinline__ void lanecnt_store(uint32_t* row, const uint32_t (&d)[NDW]) {
  if constexpr (I + 4 <= NDW) {
    v4w x; x.x = d[I]; x.y = d[I + 1]; x.z = d[I + 2]; x.w = d[I + 3];
    *reinterpret_cast<RH_GLOBAL v4wu*>(reinterpret_cast<uintptr_t>(row + I)) = x;
    lanecnt_store<NDW, I + 4>(row, d);
  } else if constexpr (I + 2 <= NDW) {
    *reinterpret_cast<RH_GLOBAL u64u*>(reinterpret_cast<uintptr_t>(row + I)) = ((uint64_t)d[I + 1] << 32) | d[I];
    lanecnt_store<NDW, I + 2>(row, d);
  } else if constexpr (I < NDW) {
    *reinterpret_cast<RH_GLOBAL uint32_t*>(reinterpret_cast<uintptr_t>(row + I)) = d[I];
  }
}
template <int NDW, int I = 0>
__device__ __forceinline__ void lanecnt_load(const uint32_t* row, uint32_t (&d)[NDW]) {
  if constexpr (I + 4 <= NDW) {
    const v4w x = *reinterpret_cast<const RH_GLOBAL v4wu*>(reinterpret_cast<uintptr_t>(row + I));
    d[I] = x.x; d[I + 1] = x.y; d[I + 2] = x.z; d[I + 3] = x.w;
    lanecnt_load<NDW, I + 4>(row, d);
  } else if constexpr (I + 2 <= NDW) {
    const uint64_t x = *reinterpret_cast<const RH_GLOBAL u64u*>(reinterpret_cast<uintptr_t>(row + I));
    d[I] = (uint32_t)x; d[I + 1] = (uint32_t)(x >> 32);
    lanecnt_load<NDW, I + 2>(row, d);
  } else if constexpr (I < NDW) {
    d[I] = *reinterpret_cast<const RH_GLOBAL uint32_t*>(reinterpret_cast<uintptr_t>(row + I));
  }
}

// LDS in front of the window: wtot[NW][KP] (KP = K rounded up to 4: a wavefront's totals are one contiguous, 16-byte aligned row) | gbx[KP + 128] | nullcnt[NNODES] | nullw[NNODES][NW] | misc[4] | bm[NBM][kBmWords] | dtab[NW][kDenseCap] (schemas with
// a dense list) | bmw0[NB0][NW] u64   (host mirror: spec_lds_fixed_words_host)
__host__ __device__ constexpr uint32_t spec_lds_fixed_words(int K, int nnodes, int nw, int nbm, int ndense, int nb0) {
  return (((uint32_t)(K > 0 ? K : 1) + 3) & ~3u) * (uint32_t)(nw + 1) + 128u + (uint32_t)((nnodes + 3) & ~3) + (uint32_t)((nnodes * nw + 3) & ~3) + 4 + (uint32_t)(nbm * kBmWords) +
         (ndense > 0 ? (uint32_t)(nw * kDenseCap) : 0u) + (((uint32_t)(nb0 * nw * 2) + 3) & ~3u);
}

// Tile geometry of a specialised kernel: S::TILE records = S::TILE threads = NW wavefronts per workgroup.
template <class S>
struct TileOf {
  static constexpr int T = S::TILE;
  static constexpr int NW = S::TILE / 64;
  static_assert(S::TILE % 64 == 0 && S::TILE >= 64 && S::TILE <= 1024, "tile must be 1..16 wavefronts");
};

template <class S>
struct SpecSmem {
  uint32_t* wtot;
  uint32_t* gbx;       // [KP + 128] single-pass form: the tile's chunk-relative base per counter, from the look-back wave to every wave; its exchange area
  uint32_t* nullcnt;
  uint32_t* nullw;
  uint32_t* misc;
  uint32_t* bm;
  uint32_t* dtab;
  uint64_t* bmw0;
  uint8_t* win;
  __device__ __forceinline__ SpecSmem(const KParams& P, uint8_t* smem) {
    uint32_t* p = reinterpret_cast<uint32_t*>(smem);
    wtot = p; p += (((S::K > 0 ? S::K : 1) + 3) & ~3) * (S::TILE / 64);
    gbx = p; p += (((S::K > 0 ? S::K : 1) + 3) & ~3) + 128;      // + the look-back wave's exchange area [64][2]
    nullcnt = p; p += ((S::NNODES + 3) & ~3);
    nullw = p; p += ((S::NNODES * (S::TILE / 64) + 3) & ~3);
    misc = p; p += 4;
    bm = p; p += S::NBM * kBmWords;                   // a multiple of 16 bytes
    dtab = p; p += S::NDENSE > 0 ? (S::TILE / 64) * kDenseCap : 0;
    bmw0 = reinterpret_cast<uint64_t*>(p); p += (S::NB0 * (S::TILE / 64) * 2 + 3) & ~3;   // (8-byte aligned: every part is a multiple of 16 bytes)
    win = reinterpret_cast<uint8_t*>(p);
  }
};

// --------------------------------------------------------------------------
// Item-dense handling of a top-level array / map without nested lists, in the emit kernel's trusted fast walk.
//
// One lane = one record leaves the block loop of a list running for as many iterations as the LONGEST list of the wave,
// with every store of an iteration (item offsets, string bytes) paying for 64 lanes whatever the number that still has an
// item (tools/storecost.hip) -- at 0..3 items per record half of the lanes of an average iteration are idle.  Here the
// record lanes only FIND the items (phase A: block headers + a counters-only skip of the item body, the window position
// of every item goes to a per-wave LDS table in row order), and the items are then materialised one lane = one ITEM
// (phase B: 64 consecutive rows of the child domain per step, every lane busy, every store of the step covering one dense
// span of its column).  An item's byte offsets inside the string columns of the body are the wave scan of the items'
// sizes (a second counters-only walk of the body, now one lane per item) on top of the wave's base.  The table holds
// kDenseCap items; a wave with more alternates the two phases in rounds.  Only for tiles the size pass cleared (no
// anomaly anywhere: RH_TRUST), so neither phase carries an error path; every other tile takes the per-record loop.
// --------------------------------------------------------------------------
// --------------------------------------------------------------------------
// Wavefront-parallel item scan (round 6) for the list of ONE record larger than the window -- an array of millions of strings.
// Finding the items of such a list is a chain of dependent reads (an item's start is the end of the one before it): one lane
// alone pays a full LDS round trip and ~80 issue slots per item (~500 cycles; rounds 1-5: ~1400, from global memory), while 63
// lanes watch.  Here every lane SPECULATES: lane j walks the item body as if an item began at byte cur + j (and cur + 64 + j, ...) and
// notes where that item would end -- one step of the body for 256 candidate starts at once, reads that never leave the window's
// guard (SlideSrc) and touch no counter but their own copy -- and the true chain is then followed through those 128 answers
// with v_readlane, a few cycles per item, until it leaves the 256 bytes.  Block headers (one per block of items) are read in
// between, wave-uniformly.  SIZE = false: the emit kernel's phase A (dense_list: positions -> tab, the bytes are trusted);
// SIZE = true: the size walk of the ranged kernel (giant_list_size: counts and byte totals, every anomaly marks the lane `redo`).
// --------------------------------------------------------------------------
template <class S, int LID, int D, int DEPTH, bool SIZE, class Src, class Body>
__device__ __forceinline__ void item_scan(const SCtx<S>& c, const Src& src, Lane& L, Body&& body, int owner, uint32_t wbase, uint32_t lo, uint32_t limit,
                                          uint32_t* tab, bool& inlist, uint32_t& top) {
  uint32_t& rm = c.remaining(DEPTH);
  // the list's state, wave-uniform (the owner lane's)
  uint32_t cur_s = (uint32_t)__builtin_amdgcn_readlane((int)L.cur, owner);
  uint32_t rm_s = (uint32_t)__builtin_amdgcn_readlane((int)rm, owner);
  uint32_t idx_s = (uint32_t)__builtin_amdgcn_readlane((int)c.cnt[D], owner) - wbase;
  bool in_s = ((__ballot(inlist) >> owner) & 1ull) != 0;
  bool bad_s = false;
  uint32_t acc[SCtx<S>::KL1];                               // SIZE: bytes the items met so far add to the body's string columns
  static_for<0, S::KL>([&](auto ik) { acc[decltype(ik)::value] = 0; });
  while (in_s && idx_s < limit && !bad_s) {
    if (rm_s == 0) {                                        // at a block boundary: count, or the 0 terminator
      uint32_t raw, n;
      const bool ok = varint32(src.ld4(cur_s), SIZE ? (uint32_t)__builtin_amdgcn_readlane((int)L.end, owner) - cur_s : 4u, raw, n);      // (a cursor past the end: avail wraps, the end-of-record check of the fast walk catches it)
      raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw);
      n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
      if (SIZE && (!__builtin_amdgcn_readfirstlane((int)ok) || (raw & 1u))) { bad_s = true; break; }     // a long / negative count: the careful walk's
      cur_s += n;
      if ((raw >> 1) == 0) { in_s = false; break; }
      rm_s = raw >> 1;
    }
    // the item body from kScanSets x 64 candidate starts
    uint32_t nxt[kScanSets];
    uint64_t redm[kScanSets];
    SCtx<S> cz[kScanSets] = {c, c, c, c};
    static_assert(kScanSets == 4, "cz is initialised for four sets");
    // A list of plain strings (S::dense_str): a candidate is one length varint -- decoded by hand from unchecked window reads,
    // the four sets' reads in flight together, instead of four passes through the string handler (its in-window tests, its
    // validity and counter plumbing: ~3,000 cycles per 256-byte span, 380 per 28-byte item of a 2M-item array; profiles/r06_y_*).
    // Only while the span and a head behind it lie inside the staged bytes; the handler takes the steps near the window's end.
    bool byhand = false;
    uint32_t padv = 0, pnn = 0;      // (by hand) a candidate's advance / length-varint bytes, one byte per set; advance 0: not a plain short item
    if constexpr (S::dense_str(LID) >= 0 && Src::kMoves) {
      byhand = __builtin_amdgcn_readfirstlane((int)((cur_s - src.wa) + (uint32_t)(64 * kScanSets) + 8u <= src.wlen)) != 0;
      if (byhand) {
        uint32_t y[kScanSets];
#pragma unroll
        for (int h = 0; h < kScanSets; h++) y[h] = LdsAbsSrc().ld4(cur_s + (uint32_t)h * 64u + c.lane);
#pragma unroll
        for (int h = 0; h < kScanSets; h++) {
          uint32_t raw, n;
          const bool ok = varint24(y[h], 4u, raw, n);
          const uint32_t len = raw >> 1;
          static_for<0, S::KL>([&](auto ik) { constexpr int k = decltype(ik)::value; cz[h].cnt[k] = k == S::dense_str(LID) ? len : 0u; });
          nxt[h] = (uint32_t)h * 64u + c.lane + n + len;
          redm[h] = SIZE ? __ballot(!ok || (raw & 1u) != 0u) : 0ull;      // (a longer or a negative length: the careful walk's)
          const bool plain = ok && (raw & 1u) == 0u && n + len < 256u;
          padv |= (plain ? n + len : 0u) << (8 * h);
          pnn |= (plain ? n : 0u) << (8 * h);
        }
      }
    }
    if (!byhand) {
#pragma unroll
    for (int h = 0; h < kScanSets; h++) {
      static_for<0, S::KL>([&](auto ik) { constexpr int k = decltype(ik)::value; cz[h].cnt[k] = 0; });
      Lane Lc;
      Lc.live = true; Lc.pres = true; Lc.err = 0; Lc.edetail = 0; Lc.redo = false;
      Lc.pstk = 0; Lc.lstk = 0; Lc.sstk = 0; Lc.la = 0;
      Lc.cur = cur_s + (uint32_t)h * 64u + c.lane;
      Lc.end = 0xFFFFFFFFu;
      if constexpr (SIZE) body(IC<0>{}, cz[h], Lc);
      else body(IC<0>{}, SkipCtx<SCtx<S>>(cz[h]), Lc);
      nxt[h] = Lc.cur - cur_s;
      redm[h] = SIZE ? __ballot(Lc.redo) : 0ull;
    }
    }
    // follow the chain through the candidates (a branch per set: reading all four sets' answers for a hop and selecting among
    // them with scalar selects was measured slower in the size walk -- eight and more v_readlane per hop: 1.54 -> 2.06 ms for four
    // 9,000-item arrays -- and 9 % faster in the emit walk, profiles/r06_y_*; not kept)
    uint32_t pos = 0;
    // The hops of a list of plain strings (by hand): a candidate's answer is ONE byte of a packed register -- its advance, the set
    // picked by a shift -- so a hop is one v_readlane (two in the size walk, which also needs the bytes of the length varints: the
    // items' string bytes are the distance hopped minus those) and a dozen scalar instructions instead of a branch per set, the
    // anomaly masks and a counter read per hop below (~45 instructions per item: 1.54 ms per 9,000-item array and pass,
    // profiles/r06_y_*).  A candidate that is not a plain item of fewer than 256 bytes has advance 0 and is left to the loop below.
    if constexpr (S::dense_str(LID) >= 0 && Src::kMoves) {
      if (byhand) {
        uint32_t sumn = 0, hops = 0;
        const uint32_t room = limit - idx_s;                               // (> 0: the loop around this)
        const uint32_t left = rm_s < room ? rm_s : room;                   // hops this block and this round allow
        while (pos < (uint32_t)(64 * kScanSets) && hops < left) {
          const int pl = (int)(pos & 63u);
          const uint32_t sh = (pos >> 3) & 24u;
          const uint32_t a = ((uint32_t)__builtin_amdgcn_readlane((int)padv, pl) >> sh) & 0xFFu;
          if (a == 0u) break;
          if constexpr (SIZE) sumn += ((uint32_t)__builtin_amdgcn_readlane((int)pnn, pl) >> sh) & 0xFFu;
          else { if (c.lane == 0) tab[idx_s - lo + hops] = cur_s + pos; }
          pos += a;
          hops++;
        }
        rm_s -= hops; idx_s += hops;
        if constexpr (SIZE) acc[S::dense_str(LID)] += pos - sumn;
      }
    }
    while (pos < (uint32_t)(64 * kScanSets) && rm_s > 0 && idx_s < limit) {
      const int pl = (int)(pos & 63u), ps = (int)(pos >> 6);
      if constexpr (!SIZE) {
        if (c.lane == 0) tab[idx_s - lo] = cur_s + pos;
      }
      uint32_t nx = 0;
#pragma unroll
      for (int h = 0; h < kScanSets; h++) {
        if (ps == h) {                                      // (wave-uniform)
          if constexpr (SIZE) {
            if ((redm[h] >> pl) & 1ull) bad_s = true;
            static_for<0, S::KL>([&](auto ik) {
              constexpr int k = decltype(ik)::value;
              if constexpr (S::dense_body(LID, k)) acc[k] += (uint32_t)__builtin_amdgcn_readlane((int)cz[h].cnt[k], pl);
            });
          }
          nx = (uint32_t)__builtin_amdgcn_readlane((int)nxt[h], pl);
        }
      }
      if (bad_s) break;
      pos = nx;
      rm_s--; idx_s++;
    }
    if (bad_s) break;
    cur_s += pos;
    if constexpr (SIZE && Src::kSlide) {                    // keep the window under the cursor (no position is noted anywhere here)
      const uint32_t delta = src.advance_to(cur_s, c.lane, owner);
      cur_s -= delta; L.cur -= delta; L.end -= delta;
    }
  }
  // back to the owner lane
  const bool mine = (int)c.lane == owner;
  if (mine) {
    L.cur = cur_s;
    rm = rm_s;
    c.cnt[D] = wbase + idx_s;
    if constexpr (SIZE) {
      static_for<0, S::KL>([&](auto ik) { constexpr int k = decltype(ik)::value; if constexpr (S::dense_body(LID, k)) c.cnt[k] += acc[k]; });
      if (bad_s) L.redo = true;
    }
  }
  inlist = mine ? in_s : false;
  top = idx_s;
}

// The size walk's side of it: the list of a sliding record (one live lane), sized by item_scan in slices of 2^16 items with the
// window moved along in between; an anomaly anywhere marks the lane `redo` (the careful walk then takes the record).  Leaves the
// lane not live, so that the block loop behind it finds nothing to do.  Any other situation: not touched.
template <class S, int LID, int D, int DEPTH, class Src, class Body>
__device__ __forceinline__ void giant_list_size(const SCtx<S>& c, const Src& src, Lane& L, Body&& body) {
  if constexpr (Src::kSlide) {
    if (!src.sliding) return;
    const uint64_t lv = __ballot(L.live);
    if (lv == 0 || (lv & (lv - 1)) != 0) return;
    const int owner = (int)__builtin_ctzll(lv);
    const uint32_t wbase = (uint32_t)__builtin_amdgcn_readlane((int)c.cnt[D], owner);
    bool inlist = L.live;
    uint32_t top = 0;
    for (uint32_t lo = 0;; lo += 65536u) {
      L.live = inlist;
      src.refill(L, c.lane);
      item_scan<S, LID, D, DEPTH, true>(c, src, L, body, owner, wbase, lo, lo + 65536u, (uint32_t*)nullptr, inlist, top);
      if (!__any(inlist) || __any(L.redo)) break;
    }
    L.live = false; L.pres = false;
  }
}

// LID: index of the list among the schema's dense lists; D: counter of its child row domain; DEPTH: its nesting depth (0).
template <class S, int LID, int D, int DEPTH, class Src, class Body>
__device__ __forceinline__ void dense_list(const SCtx<S>& c, const Src& src, Lane& L, Body&& body) {
  constexpr int NW = S::TILE / 64;
  uint32_t* const tab = c.dtab;
  // Over a SlideSrc the wavefront walks only the records of ONE range of its tile (ranged_tile): the items are those of the
  // lanes that are live here -- rows [first live lane's prefix, ...) -- and their number is only known once phase A has met
  // them (`top`), so the rounds end when no lane has an item left instead of at the wavefront's total.
  int first = 0;
  if constexpr (Src::kSlide) {
    const uint64_t lv = __ballot(L.live);
    if (lv == 0) { L.live = false; L.pres = false; return; }
    first = (int)__builtin_ctzll(lv);
  }
  const uint32_t n_w = Src::kSlide ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)c.wtot_w[D]);   // items of this wavefront
  const uint32_t wbase = (uint32_t)__builtin_amdgcn_readlane((int)c.cnt[D], first);       // tile-local row of its first item
  uint32_t carry[SCtx<S>::KL1];                              // running tile-local byte offset of the body's string columns
  static_for<0, S::KL>([&](auto ik) {
    constexpr int k = decltype(ik)::value;
    if constexpr (S::dense_body(LID, k)) carry[k] = (uint32_t)__builtin_amdgcn_readlane((int)c.cnt[k], first);
  });
  uint32_t& rm = c.remaining(DEPTH);
  bool inlist = L.live;                                      // h_list_begin: the row carries a list
  for (uint32_t lo = 0;; lo += kDenseCap) {
    const uint32_t limit = lo + kDenseCap;
    uint32_t top = 0;                                        // (SlideSrc) one past the last item this lane noted in this round
    if constexpr (Src::kSlide) { L.live = inlist; src.refill(L, c.lane); }      // (the table is empty here: positions may move)
    // phase A, one lane = one record: note where the items [lo, limit) of the wave start
#ifndef RH_V_NOSCAN
    if constexpr (Src::kSlide) {
      if (src.sliding) {      // ONE record larger than the window: its one lane would find the items at one dependent read each
        item_scan<S, LID, D, DEPTH, false>(c, src, L, body, first, wbase, lo, limit, tab, inlist, top);
        goto scanned;
      }
    }
#endif
    for (;;) {
      const bool need = inlist && rm == 0;                   // at a block boundary: count, or the 0 terminator
      uint32_t raw, n;
      (void)varint32(src.ld4(L.cur), 4u, raw, n);
      if (need) {
        L.cur += n;
        if ((raw >> 1) == 0) inlist = false;
        else rm = raw >> 1;
      }
      const uint32_t idx = c.cnt[D] - wbase;                 // row of this lane's next item, relative to the wave's first
      const bool go = inlist && idx < limit;
      if (!__any(go)) break;
      if (go) { tab[idx - lo] = L.cur; top = idx + 1; }
      L.live = go; L.pres = go;
      body(IC<0>{}, SkipCtx<SCtx<S>>(c), L);
      if (go) { rm -= 1; c.cnt[D] += 1; }
    }
  scanned:
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // a wave's DS instructions execute in order: no barrier
    // phase B, one lane = one item
    uint32_t nr;
    if constexpr (Src::kSlide) {
      uint32_t t = top;                                      // the largest `top` of the wavefront: rows are dealt in lane order
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)t, d, 64); t = o > t ? o : t; }
      nr = t > lo ? t - lo : 0u;
    } else {
      nr = n_w > lo ? (n_w - lo < (uint32_t)kDenseCap ? n_w - lo : (uint32_t)kDenseCap) : 0u;
    }
    for (uint32_t g = 0; g < nr; g += 64) {
      const uint32_t i = g + c.lane;
      const bool act = i < nr;
      Lane Li;
      Li.live = act; Li.pres = act; Li.err = 0; Li.edetail = 0; Li.redo = false;
      Li.pstk = 0; Li.lstk = 0; Li.sstk = 0; Li.la = 0;
      Li.cur = act ? tab[i] : Src::kSlide ? tab[0] : 0u;      // (an idle lane reads somewhere harmless: LDS address 0, or the round's first item)
      Li.end = 0xFFFFFFFFu;
      SCtx<S> ci = c;
      ci.cnt[D] = wbase + lo + i;
      {   // item sizes -> offsets of every item inside the body's string columns
        SCtx<S> cz = c;
        static_for<0, S::KL>([&](auto ik) {
          constexpr int k = decltype(ik)::value;
          if constexpr (S::dense_body(LID, k)) cz.cnt[k] = 0;
        });
        Lane Lz = Li;
        body(IC<0>{}, SkipCtx<SCtx<S>>(cz), Lz);
        static_for<0, S::KL>([&](auto ik) {
          constexpr int k = decltype(ik)::value;
          if constexpr (S::dense_body(LID, k)) {
            const uint32_t d = cz.cnt[k];
            const uint32_t incl = wave_incl_scan(d, c.lane);
            ci.cnt[k] = carry[k] + incl - d;
            carry[k] += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          }
        });
      }
      body(IC<1>{}, ci, Li);
    }
    if constexpr (Src::kSlide) { if (!__any(inlist)) break; }
    else if (limit >= n_w) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  L.live = false; L.pres = false;                            // (h_list_end restores both from the stacks)
}

// The walk of a tile whose bytes are staged in the LDS window in one piece (`fits`): cursors become LDS byte addresses (walk.h
// LdsAbsSrc: k_emit -3 %, profiles/r03ap_abs_window_ab.txt).  !fits: the careful walk straight from global memory, one dependent
// HBM round trip per head (rounds 1-5's only answer to a tile past the window; profiles/r06_a_*: 6.3 ms where the staged walk
// takes 1) -- left to the single-pass form; the two-pass kernels give such tiles to the RANGED pair (rh_spec_size_r /
// rh_spec_emit_r below) and never instantiate it.
template <class S, bool EMIT, bool CAREFUL>
__device__ __forceinline__ void spec_run_walk(const KParams& P, const SCtx<S>& c, uint8_t* win, Lane& L, bool fits, uint64_t wb16) {
  if (fits) {
    const uint32_t wa = (uint32_t)(uintptr_t)(RH_LDS uint8_t*)win;
    L.cur += wa; L.end += wa;
    LdsAbsSrc src;
    S::template walk<EMIT, CAREFUL>(c, src, L);
  } else {
    GlobalSrc src{P.data + wb16, P.data_len - wb16};
    S::template walk<EMIT, true>(c, src, L);
  }
}

// --------------------------------------------------------------------------
// Tiles past the LDS window (round 6; VERDICT round 5, item 3: they were walked from global memory by the careful form, one
// dependent HBM round trip per head -- 38 % of the tiles of a skewed workload, 20 ms where the uniform one takes 1).
//
// Such a tile is processed in RANGES of consecutive records, split by offsets[]: range [a, b) = the longest run of records
// from `a` whose bytes fit the window.  It is staged by the whole workgroup like a tile's window and walked -- fast forms,
// trust, item-dense lists and all -- by the lanes that own its records; the other lanes stand by, and a wavefront without a
// record of the range skips the walk altogether, so every wavefront walks about once whatever the number of ranges.  A SINGLE
// record larger than the window is a range of its own, `sliding` (walk.h SlideSrc): its first bytes are staged by its own
// wavefront and the window follows its cursor from list iteration to list iteration, the items of its top-level lists still
// emitted one lane per item (dense_list) -- what a record with an array of millions of items needs.
// `f(src, inr, rb16)` is called once per range by every wavefront that owns a record of it (all 64 lanes; inr = this lane's
// record is in the range; rb16 = payload offset of window byte 0).  Contains workgroup barriers: uniform control flow.
// --------------------------------------------------------------------------
template <class S, class F>
__device__ __forceinline__ void ranged_tile(const KParams& P, const SpecSmem<S>& s, const Geo& g, uint64_t o1, uint32_t tid, F&& f) {
  constexpr int T = TileOf<S>::T, NW = TileOf<S>::NW;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t wa = (uint32_t)(uintptr_t)(RH_LDS uint8_t*)s.win;
  const uint32_t wcap = P.win_bytes & ~15u;
  // A range is staged WITH the kRangeSlack bytes behind its last record (the next record's first bytes, or zeros at the end of
  // the payload): the walk's window reads reach up to 20 bytes past a cursor, and a read that is not completely inside the staged
  // bytes is served from global memory (SlideSrc) -- for the last fields of a range's last record that was a global load behind
  // every store the wavefront had in flight (vmcnt counts in order on this part): the friendly workload forced through the
  // ranged pair took the emit kernel 5.2 ms where it takes the kernel of the tiles that fit 0.73 ms (profiles/r06_k_*).
  const uint32_t wfit = wcap > 4096u ? wcap - kRangeSlack : wcap;
  uint32_t* const rng = s.gbx + SCtx<S>::KP;           // [NW] records of the range per wavefront (the single-pass form's exchange area: unused here)
  // Records so large that a range holds only a few of them (a 200-column record is 1.5 KB: a dozen records per 17 KB window)
  // leave most lanes of every range idle; such a tile is walked DIRECTLY instead: every lane on its own
  // record, every read served from global memory (a SlideSrc with nothing staged) -- sixteen resident wavefronts per CU hide
  // those round trips better than eleven busy lanes use a staged window (measured on the 200-column workload, 1M records:
  // ranges 21.3 ms, the interpreter's global walk 9.8 ms, profiles/r06_e_*).  A record that is larger than the window by
  // itself still gets its sliding range, in its place in the order of the records (the wave counters of a wide schema are
  // scanned range by range): the tile is then [records in front of it, direct] [it, sliding] [records behind it, direct] ...
  if constexpr (SCtx<S>::kWaveCtr) {
    // A wide schema's tile is ONE wavefront (program.h kWideTile): everything about its ranges is wave-local, and `f` has ONE
    // call site -- with several, the walk (tens of thousands of instructions) stays a function of its own whose captured
    // context lives in scratch memory: 38 scratch accesses per column of the 200-column workload (profiles/r06_u_*).
    static_assert(NW == 1, "a wide schema's tile is one wavefront");
    const uint64_t tb = P.offsets[g.rec0], te = P.offsets[g.rec0 + g.nrec];
    const uint64_t o0m = tid < g.nrec ? P.offsets[g.rec0 + tid] : te;
    const uint32_t o0lo = (uint32_t)o0m, o0hi = (uint32_t)(o0m >> 32), o1lo = (uint32_t)o1, o1hi = (uint32_t)(o1 >> 32);
    auto lane64 = [&](uint32_t lo, uint32_t hi, int l) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, l); };
    const bool bigme = tid < g.nrec && (o1 - (o0m & ~15ull)) > (uint64_t)wfit;
    const uint64_t bigm = __ballot(bigme);
    uint64_t bigbytes = 0;
    for (uint64_t m = bigm; m; m &= m - 1) { const int l = (int)__builtin_ctzll(m); bigbytes += lane64(o1lo, o1hi, l) - lane64(o0lo, o0hi, l); }
    // (fewer than kDirectLanes records per range on average -- bytes / window ranges for nrec records: the 200-column workload has 11)
    const bool direct = ((te - (tb & ~15ull)) - bigbytes) * (uint64_t)kDirectLanes > (uint64_t)g.nrec * wcap;
    uint32_t a = 0;
    while (a < g.nrec) {                               // (wave-uniform)
      const uint64_t oa = lane64(o0lo, o0hi, (int)a);  // first byte of record `a`
      SlideSrc src{wa, 0u, wcap, P.data, 0ull, false};
      uint64_t rb16 = oa & ~15ull;
      uint32_t b;
      bool single;
      if (direct) {
        single = ((bigm >> a) & 1ull) != 0;
        if (single) {
          b = a + 1u;
        } else {                                       // the records up to the next giant (or the end of the tile), every lane on its own
          const uint64_t m = bigm & (~0ull << a);
          b = m ? (uint32_t)__builtin_ctzll(m) : g.nrec;
          rb16 = tb & ~15ull;
          src.g = P.data + rb16; src.glim = P.data_len - rb16;
#if defined(RH_WIDE_SCHEMA) && !defined(RH_V_NOLANES)
          // lane windows (walk.h SlideSrc::lanes): every lane stages its own record through a slice of the window
          const uint32_t stride = (wcap / 64u) & ~15u;
          if (stride >= 96u) {
            const bool mine = tid >= a && tid < b;
            src.lanes = true; src.lwin = stride - 16u; src.lw = wa + lane * stride;
            src.p0 = wa + (mine ? (uint32_t)((o0m & ~15ull) - rb16) : 0u);
            if (mine) src.stage_lane();
          }
#endif
        }
      } else {
        const bool fitme = tid >= a && tid < g.nrec && (o1 - rb16) <= (uint64_t)wfit;      // offsets are monotonic: a prefix of [a, nrec)
        const uint32_t cnt = (uint32_t)__popcll(__ballot(fitme));
        single = cnt == 0;                             // record `a` alone is larger than the window
        b = single ? a + 1u : a + cnt;
        if (!single) {
          uint64_t re = lane64(o1lo, o1hi, (int)(b - 1u)) + (wfit < wcap ? (uint64_t)kRangeSlack - 16u : 0u);
          if (re > P.data_len + 16u) re = P.data_len + 16u;       // (stage_window zero-fills one vector past the payload)
          stage_window<T>(P, s.win, rb16, re, tid);
          src.wlen = (uint32_t)((re - rb16 + 15) & ~15ull);
          src.g = P.data + rb16; src.glim = P.data_len - rb16;
        }
      }
      if (single) {                                    // a sliding range: the window follows the record's cursor
        const uint64_t left = P.data_len - rb16 + 15ull;
        src.wlen = left < (uint64_t)wcap ? (uint32_t)(left & ~15ull) : wcap;
        src.g = P.data + rb16; src.glim = P.data_len - rb16; src.sliding = true;
        stage_wave(src.g, src.glim, wa, src.wlen, lane);
      }
      __syncthreads();
      f(src, tid >= a && tid < b, rb16);
      __syncthreads();                                 // the window is free for the next range
      a = b;
    }
    return;
  }
  // (a range's bounds come from the record bounds the lanes hold in registers, through LDS -- not from offsets[] again: two
  //  dependent global loads per range in front of its staging)
  unsigned long long* const omax = reinterpret_cast<unsigned long long*>(rng + 32);      // [NW] end of the wavefront's last record that fits
  unsigned long long* const o1a = reinterpret_cast<unsigned long long*>(rng + 48);       // end of record `a`
  const uint32_t o1lo = (uint32_t)o1, o1hi = (uint32_t)(o1 >> 32);
  auto o1_of = [&](int l) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)o1hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)o1lo, l); };
  uint32_t a = 0;
  uint64_t ra = P.offsets[g.rec0];                     // first byte of record `a`
  while (a < g.nrec) {                                 // (workgroup-uniform)
    const uint64_t rb16 = ra & ~15ull;
    const bool fitme = tid >= a && tid < g.nrec && (o1 - rb16) <= (uint64_t)wfit;      // offsets are monotonic: a prefix of [a, nrec)
    const uint64_t fm = __ballot(fitme);
    const uint32_t wc = (uint32_t)__popcll(fm);
    if (wc) { const uint64_t e = o1_of(63 - (int)__builtin_clzll(fm)); if (lane == 0) omax[wave] = e; }
    if (wave == a / 64u) { const uint64_t e = o1_of((int)(a & 63u)); if (lane == 0) *o1a = e; }
    if (lane == 0) rng[wave] = wc;
    __syncthreads();
    uint32_t cnt = 0;
    uint64_t rend = *o1a;
#pragma unroll
    for (int w = 0; w < NW; w++) { const uint32_t x = rng[w]; cnt += x; if (x) rend = omax[w]; }
    const bool single = cnt == 0;                      // record `a` alone is larger than the window
    const uint32_t b = single ? a + 1u : a + cnt;
    ra = rend;                                         // (== offsets[rec0 + b])
    uint32_t staged;
    if (!single) {
      uint64_t re = rend + (wfit < wcap ? (uint64_t)kRangeSlack - 16u : 0u);
      if (re > P.data_len + 16u) re = P.data_len + 16u;       // (stage_window zero-fills one vector past the payload)
      stage_window<T>(P, s.win, rb16, re, tid);
      staged = (uint32_t)((re - rb16 + 15) & ~15ull);
    } else {
      const uint64_t left = P.data_len - rb16 + 15ull;
      staged = left < (uint64_t)wcap ? (uint32_t)(left & ~15ull) : wcap;
      if (wave == a / 64u) stage_wave(P.data + rb16, P.data_len - rb16, wa, staged, lane);
    }
    __syncthreads();
    const bool inr = tid >= a && tid < b;
    if (__any(inr)) {
      // a range staged whole is read like the window of a tile that fits (RangeSrc: no in-window test behind every head)
      if (single) {
        const SlideSrc src{wa, staged, wcap, P.data + rb16, P.data_len - rb16, true};
        f(src, inr, rb16);
      } else {
        f(RangeSrc{}, inr, rb16);
      }
    }
    __syncthreads();                                   // the window is free for the next range
    a = b;
  }
}

// lane set-up for one range: only the lanes whose record is in it are live; cursors are LDS addresses (window byte 0 = wa)
__device__ __forceinline__ void lane_init_range(Lane& L, bool inr, uint64_t o0, uint64_t o1, uint64_t rb16, uint32_t wa) {
  L.live = inr; L.pres = inr;
  L.err = 0; L.edetail = 0; L.redo = false;
  L.pstk = 0; L.lstk = 0; L.sstk = 0; L.la = 0;
  L.cur = wa + (inr ? (uint32_t)(o0 - rb16) : 0u);
  L.end = wa + (inr ? (uint32_t)(o1 - rb16) : 0u);
  if (inr && (o1 - rb16) > 0xFFFFFFF0ull - 0x40000ull) L.err = E_EOB;      // a record beyond 32-bit cursors
}

template <class S>
__device__ __forceinline__ void spec_ctx_init(SCtx<S>& c, const KParams& P, const SpecSmem<S>& s, const Geo& g, uint32_t tid) {
  static_for<0, SCtx<S>::KL1>([&](auto ik) { constexpr int k = decltype(ik)::value; c.cnt[k] = 0; c.gb[k] = 0; });
  c.wv = s.gbx;
  static_for<0, (S::DEPTH > 0 ? S::DEPTH : 1)>([&](auto id) { c.rem[decltype(id)::value] = 0; });
  c.nullcnt = s.nullcnt; c.nullw_w = s.nullw + (tid >> 6); c.bm = s.bm; c.sym_off = P.sym_off; c.sym_data = P.sym_data;
  c.lrow = g.lrow0 + tid; c.lane = tid & 63; c.wave_live = ((tid >> 6) * 64) < g.nrec;
  c.dtab = s.dtab + (tid >> 6) * kDenseCap; c.wtot_w = s.wtot + (tid >> 6) * SCtx<S>::KP;
  c.bmw0 = s.bmw0;
}

// --------------------------------------------------------------------------
// size pass: per-workgroup counter sums (+ first malformed record)
// --------------------------------------------------------------------------
// RANGED = false: rh_spec_size, the kernel of every tile that fits the window (a tile past it is left to rh_spec_size_r when
// P.ranged says that kernel follows, else walked by the fallback above).  RANGED = true: rh_spec_size_r, which only works on the
// tiles past the window (ranged_tile) -- a code object of its own, compiled when a schema first meets such tiles, so that the
// hot kernel's registers, code size and compile time are those of the tiles that fit.
// The ranged kernels' tile of workgroup b (program.h KParams::worklist): the first kBigFront workgroups take the large tiles the
// size kernel listed, the others their usual tile unless a front workgroup has it; false: no tile for this workgroup.
__device__ __forceinline__ bool ranged_tile_of_block(const KParams& P, uint32_t b, uint32_t& tile) {
  if (!P.worklist) { tile = tile_of_block(b, P.nblocks); return true; }
  if (b < kBigFront) {
    if (b >= P.worklist[0]) return false;
    tile = P.worklist[2u + b];
    return true;
  }
  tile = tile_of_block(b - kBigFront, P.nblocks);
  const uint32_t m = P.bigmark[tile];
  return !(m != 0u && m <= kBigFront);
}

template <class S, bool RANGED = false>
__device__ __forceinline__ void spec_size(const KParams& P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SpecSmem<S> s(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  RH_MARK_INIT;
  constexpr int T = TileOf<S>::T, NW = TileOf<S>::NW, KP = SCtx<S>::KP;
  uint32_t tile = 0;
  if constexpr (RANGED) { if (!ranged_tile_of_block(P, blockIdx.x, tile)) return; }      // (workgroup-uniform, ahead of every barrier)
  else tile = tile_of_block(blockIdx.x, P.nblocks);
  const Geo g = geometry<T>(P, tile);
  uint64_t o0 = 0, o1 = 0;   // this lane's record bounds: issued together with the window bounds
  if (tid < g.nrec) { o0 = P.offsets[g.rec0 + tid]; o1 = P.offsets[g.rec0 + tid + 1]; }
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (RANGED ? fits : !fits) {      // (workgroup-uniform, ahead of every barrier) the other kernel's tile ...
    // ... or nobody's: the ranged pair is not part of this call.  The call is refused (LF_NEED_RANGED: k_init / k_emit return at
    // once) and the host repeats it -- on the generic kernels, while the pair compiles; with the pair from then on.
    if (!RANGED && P.ranged == 0 && tid == 0) atomicOr(reinterpret_cast<uint32_t*>(P.first_bad) + 2, (uint32_t)LF_NEED_RANGED);
    if constexpr (!RANGED) {
      if (P.ranged != 0 && P.worklist && tid == 0 && (we - wb16) > P.big_tile_bytes) {      // a large tile: to the front of the ranged pair
        const uint32_t at = atomicAdd(&P.worklist[0], 1u);
        P.worklist[2u + at] = tile;
        P.bigmark[tile] = at + 1u;
      }
    }
    return;
  }
  if (fits) stage_window<T>(P, s.win, wb16, we, tid);
  if (tid == 0) { s.misc[0] = 0xFFFFFFFFu; s.misc[2] = 0; }
  if constexpr (SCtx<S>::kWaveCtr) for (int k = S::KL + (int)tid; k < S::K; k += T) s.wtot[k] = 0;      // (one wavefront per tile: its row of wtot)
  __syncthreads();
  RH_MARK(16);
  // wide schemas: a walk leaves the wavefront's sum of every wave counter in c.wv (SCtx::wave_total); the walk that counts -- the
  // fast one, or the careful one behind it -- is added to the tile's totals
  auto commit_wave_counters = [&]() {
    if constexpr (SCtx<S>::kWaveCtr) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int k = S::KL + (int)lane; k < S::K; k += 64) s.wtot[k] += s.gbx[k];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  };

  Lane L;
  SCtx<S> c;
  spec_ctx_init(c, P, s, g, tid);
  RH_MARK(17);
  bool careful = false;
  uint32_t tflag = 0;
  if constexpr (!RANGED) {
    {
      lane_init_from(L, g, o0, o1, wb16, tid);
      spec_run_walk<S, false, false>(P, c, s.win, L, true, wb16);
      L.redo = L.redo || L.cur > L.end;      // the one bounds check of the fast walk (walk.h read_head): a cursor past its record's end
      if (__any(L.redo)) {   // some record of this wave left the fast wire forms (or is malformed): walk the wave again, carefully
        careful = true;
        tflag |= (uint32_t)TF_REWALK_ONE;
        spec_ctx_init(c, P, s, g, tid);
        lane_init_from(L, g, o0, o1, wb16, tid);
        spec_run_walk<S, false, true>(P, c, s.win, L, true, wb16);
      }
    }
    commit_wave_counters();
  } else {
    // the tile does not fit the window: its records in ranges (ranged_tile), each walked like a tile of its own
    tflag = (uint32_t)(TF_OVER_WINDOW | TF_SUBTILED);
    lane_init_range(L, false, 0, 0, 0, 0);
    const uint32_t wa = (uint32_t)(uintptr_t)(RH_LDS uint8_t*)s.win;
    ranged_tile<S>(P, s, g, o1, tid, [&](auto src0, bool inr, uint64_t rb16) __attribute__((always_inline)) {      // (SlideSrc, or RangeSrc for a range staged whole)
      using SrcT = decltype(src0);
      Lane Lr;
      lane_init_range(Lr, inr, o0, o1, rb16, wa);
      const SrcT src = src0;               // (a sliding range moves ITS copy of the window along: refill)
      S::template walk<false, false>(c, src, Lr);
      Lr.redo = Lr.redo || Lr.cur > Lr.end;
      if (__any(Lr.redo)) {
        careful = true;
        if (!(tflag & (uint32_t)TF_REWALK_ONE)) tflag |= (uint32_t)TF_REWALK_ONE;      // (counted once per wavefront)
        static_for<0, S::KL>([&](auto ik) { constexpr int k = decltype(ik)::value; if (inr) c.cnt[k] = 0; });
        static_for<0, (S::DEPTH > 0 ? S::DEPTH : 1)>([&](auto id) { c.rem[decltype(id)::value] = 0; });
        lane_init_range(Lr, inr, o0, o1, rb16, wa);
        const SrcT src2 = src0;            // the record again from its first byte: a window that has moved is staged anew
        if constexpr (SrcT::kMoves) {
          if (src0.lanes) { if (inr) src2.stage_lane(); }      // (lane windows: every lane's slice again from its record's first byte)
          else if (src0.sliding && __any(src.g != src0.g)) stage_wave(src0.g, src0.glim, src0.wa, src0.wlen, lane);
        }
        S::template walk<false, true>(c, src2, Lr);
      }
      commit_wave_counters();
      if (inr) L = Lr;
    });
  }
  RH_MARK(18);

  // per-record counters -> HBM (16 bits each, record-major: lanecnt_store) so that k_emit can skip its size walk
  // (the ranged kernel hands them over with 32 bits each -- a record of a tile past the window is often larger than 16 bits
  //  count, and a saturated tile costs the emit kernel a careful size walk: 80 % of a giant record's emit time before)
  if constexpr (RANGED) {
    if constexpr (S::KL > 0) lanecnt_store<SCtx<S>::KL1>(P.lanecnt32 + ((size_t)tile * T + tid) * S::KL, c.cnt);      // (16-byte pieces)
    static_for<0, S::KL>([&](auto ik) {
      constexpr int k = decltype(ik)::value;
      const uint32_t v = wave_sum(c.cnt[k]);
      if (lane == 0) s.wtot[wave * KP + k] = v;
    });
    if (lane == 0 && (careful || tflag)) {
      atomicOr(&s.misc[2], (careful ? (uint32_t)TF_CAREFUL : 0u) | (tflag & (uint32_t)(TF_OVER_WINDOW | TF_SUBTILED)));
      if (tflag & (uint32_t)TF_REWALK_ONE) atomicAdd(&s.misc[2], (uint32_t)TF_REWALK_ONE);
    }
    report_errors(P, s.misc, L, g, tid, tile);
    if (tid == 0) P.tileflag[tile] = s.misc[2];
    for (int k = tid; k < S::K; k += T) {
      uint32_t tsum = 0;
      for (int w = 0; w < NW; w++) tsum += s.wtot[w * KP + k];
      P.blocksum[(size_t)k * P.nblocks + tile] = tsum;
    }
    return;
  }
  bool sat = false;
  constexpr int NDW = (S::KL + 1) / 2;
  uint32_t packed[NDW > 0 ? NDW : 1] = {};
  // No per-counter clamp (a counter beyond 16 bits flags the tile and k_emit sizes it again: the packed
  // values of a flagged tile are never read), and when every counter of the wave is below 1024 the wave totals are
  // reduced two counters per dword (64 x 1023 < 2^16: no carry between the halves) -- half the DPP reductions (k_size -2.6 %, profiles/r03ad_variants_ab.txt)
  uint32_t allor = 0;
  static_for<0, S::KL>([&](auto ik) {
    constexpr int k = decltype(ik)::value;
    const uint32_t cv = c.cnt[k];
    allor |= cv;
    packed[k / 2] = (k & 1) ? (packed[k / 2] | (cv << 16)) : cv;
  });
  sat = allor > 0xFFFFu;
  if (!__any(allor > 1023u)) {
    static_for<0, NDW>([&](auto id) {
      constexpr int d = decltype(id)::value;
      const uint32_t v = wave_sum(packed[d]);
      if (lane == 0) {
        s.wtot[wave * KP + 2 * d] = v & 0xFFFFu;
        if constexpr (2 * d + 1 < S::KL) s.wtot[wave * KP + 2 * d + 1] = v >> 16;
      }
    });
  } else {
    static_for<0, S::KL>([&](auto ik) {
      constexpr int k = decltype(ik)::value;
      const uint32_t v = wave_sum(c.cnt[k]);
      if (lane == 0) s.wtot[wave * KP + k] = v;
    });
  }
  if constexpr (S::KL > 0) lanecnt_store<NDW>(P.lanecnt + ((size_t)tile * T + tid) * NDW, packed);
  const bool anysat = __any(sat);
  // (TileFlag: the re-walk count is ADDED, one per wavefront; the other bits are the same for every wavefront that sets them)
  if (lane == 0 && (anysat || careful || tflag)) {
    atomicOr(&s.misc[2], (anysat ? (uint32_t)TF_SATURATED : 0u) | (careful ? (uint32_t)TF_CAREFUL : 0u) | (tflag & (uint32_t)(TF_OVER_WINDOW | TF_SUBTILED)));
    if (tflag & (uint32_t)TF_REWALK_ONE) atomicAdd(&s.misc[2], (uint32_t)TF_REWALK_ONE);
  }
  report_errors(P, s.misc, L, g, tid, tile);   // contains the barrier that publishes wtot and misc[2]
  if (tid == 0) P.tileflag[tile] = s.misc[2];
  for (int k = tid; k < S::K; k += T) {
    uint32_t tsum = 0;
    for (int w = 0; w < NW; w++) tsum += s.wtot[w * KP + k];
    P.blocksum[(size_t)k * P.nblocks + tile] = tsum;
  }
  RH_MARK(19);
  RH_MARK_FLUSH;
}

// --------------------------------------------------------------------------
// emit pass: re-size, scan inside the workgroup, materialise
// --------------------------------------------------------------------------
template <class S, bool RANGED = false>      // (RANGED: rh_spec_emit_r, the tiles past the window -- see spec_size)
__device__ __forceinline__ void spec_emit(const KParams& P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // the size pass found a malformed record, or the layout kernel refused (program.h LayoutFlag): nothing to emit.
  // (Both words are final before this kernel starts, so every wave of a workgroup takes the same side: without a size
  //  pass -- K == 0 -- this kernel is the writer of first_bad and does not look at it.)
  if ((S::K > 0 && P.first_bad[0] != 0) || reinterpret_cast<const uint32_t*>(P.first_bad)[2] != 0) return;
  const SpecSmem<S> s(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  RH_MARK_INIT;
  constexpr int T = TileOf<S>::T, NW = TileOf<S>::NW, KP = SCtx<S>::KP;
  uint32_t tile = 0;
  if constexpr (RANGED) { if (!ranged_tile_of_block(P, blockIdx.x, tile)) return; }      // (workgroup-uniform, ahead of every barrier)
  else tile = tile_of_block(blockIdx.x, P.nblocks);
  const Geo g = geometry<T>(P, tile);
  uint64_t o0 = 0, o1 = 0;   // this lane's record bounds: issued together with the window bounds
  if (tid < g.nrec) { o0 = P.offsets[g.rec0 + tid]; o1 = P.offsets[g.rec0 + tid + 1]; }
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  if (RANGED == ((we - (wb & ~15ull)) <= (uint64_t)P.win_bytes)) return;      // the other kernel's tile (without that kernel the call was refused: LF_NEED_RANGED)
  Lane L;
  SCtx<S> c;
  spec_ctx_init(c, P, s, g, tid);
  // uniform per-workgroup inputs, requested before the window so their latency hides behind it
  c.bufp = (const __attribute__((address_space(4))) uint64_t*)(reinterpret_cast<uintptr_t>(P.bufptr) + (size_t)g.chunk * S::NBUF * 8);
  static_for<0, S::KL>([&](auto ik) {
    constexpr int k = decltype(ik)::value;
    c.gb[k] = P.blockbase[(size_t)k * P.nblocks + tile];
  });

  // this record's counters as k_size left them (requested with the window, so no extra round trip)
  // (all_careful: no size pass classified the tiles -- K == 0, tileflag is not even written -- or RUHVRO_HIP_NO_TRUST;
  //  a saturated counter, bit 0, still sends the tile through the re-size walk)
  const uint32_t tflag = (S::K > 0 ? P.tileflag[tile] : 0u) | (P.all_careful ? 2u : 0u);
  const uint32_t rewalk = tflag & 1u;
  const bool careful = (tflag & 2u) != 0;
  constexpr int NDW = (S::KL + 1) / 2;
  uint32_t packed[NDW > 0 ? NDW : 1] = {};      // two 16-bit counters per dword, as k_size left them (unpacked after the scan)
  if constexpr (S::KL > 0 && !RANGED) lanecnt_load<NDW>(P.lanecnt + ((size_t)tile * T + tid) * NDW, packed);
  if constexpr (RANGED && S::KL > 0) lanecnt_load<SCtx<S>::KL1>(P.lanecnt32 + ((size_t)tile * T + tid) * S::KL, c.cnt);
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  RH_MARK(0);
  if (fits) stage_window<T>(P, s.win, wb16, we, tid);
  for (int i = tid; i < S::NNODES; i += T) s.nullcnt[i] = 0;
  for (int i = tid; i < S::NNODES * NW; i += T) s.nullw[i] = 0;
  for (int i = tid; i < S::NBM * kBmWords; i += T) s.bm[i] = 0;
  if constexpr (RANGED) for (int i = tid; i < S::NB0 * NW; i += T) s.bmw0[i] = 0;      // (ranged_tile ORs a wavefront's bitmap words together)
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  __syncthreads();
  RH_MARK(1);
  const uint32_t wa = (uint32_t)(uintptr_t)(RH_LDS uint8_t*)s.win;

  if (S::KL > 0) {
    bool narrow = !rewalk && !RANGED;      // (the ranged kernel's counters are 32-bit: scanned one by one)
    if (rewalk && !RANGED) {   // a counter saturated its 16-bit slot (a very long string / list): size this tile again
      static_for<0, S::KL>([&](auto ik) { c.cnt[decltype(ik)::value] = 0; });
      if constexpr (!RANGED) {
        lane_init_from(L, g, o0, o1, wb16, tid);
        spec_run_walk<S, false, true>(P, c, s.win, L, true, wb16);
      }
    } else if constexpr (!RANGED) {
      // every counter of the wave below 1024: the inclusive scan of a PACKED dword cannot carry from its low half into
      // its high half (64 x 1023 < 2^16), so the wave scans two counters per dword -- half the DPP adds
      uint32_t allor = 0;
      static_for<0, NDW>([&](auto id) { allor |= packed[decltype(id)::value]; });
      narrow = !__any((allor & 0xFC00FC00u) != 0);
      if (!narrow) static_for<0, S::KL>([&](auto ik) {
        constexpr int k = decltype(ik)::value;
        c.cnt[k] = (k & 1) ? packed[k / 2] >> 16 : packed[k / 2] & 0xFFFFu;
      });
    }
    RH_MARK(3);
    if (narrow) {
      uint32_t tot[SCtx<S>::KLP] = {};
      static_for<0, NDW>([&](auto id) {
        constexpr int d = decltype(id)::value;
        const uint32_t v = packed[d];
        const uint32_t incl = wave_incl_scan(v, lane);
        const uint32_t ex = incl - v;
        c.cnt[2 * d] = ex & 0xFFFFu;
        tot[2 * d] = incl & 0xFFFFu;
        if constexpr (2 * d + 1 < S::KL) { c.cnt[2 * d + 1] = ex >> 16; tot[2 * d + 1] = incl >> 16; }
      });
      if (lane == 63) {      // this wavefront's totals: one contiguous row (16-byte stores)
        static_for<0, SCtx<S>::KLP / 4>([&](auto iq) {
          constexpr int q = decltype(iq)::value;
          v4w x; x.x = tot[4 * q]; x.y = tot[4 * q + 1]; x.z = tot[4 * q + 2]; x.w = tot[4 * q + 3];
          *reinterpret_cast<v4w*>(s.wtot + wave * KP + 4 * q) = x;
        });
      }
    } else {
      static_for<0, S::KL>([&](auto ik) {
        constexpr int k = decltype(ik)::value;
        const uint32_t v = c.cnt[k];
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s.wtot[wave * KP + k] = incl;
        c.cnt[k] = incl - v;
      });
    }
    RH_MARK(4);
    __syncthreads();
    RH_MARK(5);
    if constexpr (NW > 1) {
      // workgroup-local exclusive prefix.  Lane k sums counter k's totals of the wavefronts in front of this one (three
      // LDS reads at most); every lane then takes counter k's sum from lane k with v_readlane: 2 VALU per counter, not
      // the 5-6 of selecting and adding NW - 1 broadcast totals per counter in every lane
      // (v_readlane selects among 64 lanes: schemas with more counters -- up to kMaxCounters = 96 -- take them 64 at a time)
      static_for<0, (S::KL + 63) / 64>([&](auto ig) {
        constexpr int k0 = decltype(ig)::value * 64;
        constexpr int kn = S::KL - k0 < 64 ? S::KL - k0 : 64;
        uint32_t prev = 0;
        if (lane < (uint32_t)kn) {
          for (int w = 0; w < NW - 1; w++) prev += (int)wave > w ? s.wtot[w * KP + k0 + lane] : 0u;
        }
        static_for<0, kn>([&](auto ik) {
          constexpr int k = decltype(ik)::value;
          c.cnt[k0 + k] += (uint32_t)__builtin_amdgcn_readlane((int)prev, k);
        });
      });
    }
  }

  RH_MARK(6);
  // wide schemas: the tile's chunk-relative base of every wave counter -> LDS, where SCtx::wave_offset keeps it running
  // (behind the re-size walk of a saturated tile, whose wave_total calls use the same words)
  if constexpr (SCtx<S>::kWaveCtr) {
    for (int k = S::KL + (int)tid; k < S::K; k += T) s.gbx[k] = P.blockbase[(size_t)k * P.nblocks + tile];
    __syncthreads();
  }
  if constexpr (!RANGED) {
    lane_init_from(L, g, o0, o1, wb16, tid);
    if (careful) {
      spec_run_walk<S, true, true>(P, c, s.win, L, true, wb16);
    } else {
      spec_run_walk<S, true, false>(P, c, s.win, L, true, wb16);
      if (L.redo) L.err = E_INTERNAL;   // k_size walks the same bytes and would have flagged the tile
    }
  } else {
    // the tile in ranges, as the size pass walked it (ranged_tile: same offsets, same window, same ranges)
    lane_init_range(L, false, 0, 0, 0, 0);
    ranged_tile<S>(P, s, g, o1, tid, [&](const auto& src, bool inr, uint64_t rb16) __attribute__((always_inline)) {      // (inlined: a walk that stays a function keeps its captured context in scratch memory)
      Lane Lr;
      lane_init_range(Lr, inr, o0, o1, rb16, wa);
      if (careful) {
        S::template walk<true, true>(c, src, Lr);
      } else {
        S::template walk<true, false>(c, src, Lr);
        if (Lr.redo) Lr.err = E_INTERNAL;
      }
      if (inr) L = Lr;
    });
  }
  RH_MARK(7);
  report_errors(P, s.misc, L, g, tid, tile);   // barrier inside: nullcnt + staging complete
  RH_MARK(8);
  for (int i = tid; i < S::NNODES; i += T) {
    uint32_t v = s.nullcnt[i];
    for (int w = 0; w < NW; w++) v += s.nullw[i * NW + w];
    if (v) atomicAdd(&P.nullcount[((size_t)i * P.k + g.chunk) * P.null_slots + (tile & (P.null_slots - 1))], v);
  }
  if constexpr (S::NB0 > 0) {   // the tile's domain-0 bitmap words (SCtx::put_word0): one lane per word
    for (uint32_t i = tid; i < (uint32_t)(S::NB0 * NW); i += T) {
      const uint32_t slot = i / NW, w = i % NW;
      if (w * 64u < g.nrec) st_global<uint64_t, false>(c.buf(S::bm0buf(slot)), (g.lrow0 >> 6) + w, c.bmw0[i]);
    }
  }
  if constexpr (S::NBM > 0) {   // the tile's child-domain bitmap words: one atomicOr per non-zero word (the first and
    static_for<0, S::NBM>([&](auto ib) {   // the last word of a run are shared with the neighbouring tiles)
      constexpr int slot = decltype(ib)::value;
      const uint32_t w0 = c.gb[S::bmdom(slot) - 1] >> 5;
      for (int w = tid; w < kBmWords; w += T) {
        const uint32_t v = s.bm[slot * kBmWords + w];
        if (v) atomic_or_global(c.buf(S::bmbuf(slot)), (uint64_t)w0 + w, v);
      }
    });
  }

  RH_MARK(9);
  RH_MARK_FLUSH;
}

// --------------------------------------------------------------------------
// Single-pass form: size walk, scan across tiles and emit walk in ONE launch.
//
// The two-pass path reads every record twice from HBM (and moves 24 bytes of counters per record between the passes) only
// because a tile cannot place its variable-length output before it knows what all tiles in front of it produce.  Here a
// workgroup stages its window once, sizes it (the size pass's walk), publishes its K totals and LOOKS BACK over the tiles
// in front of it in its chunk for their totals / inclusive prefixes (decoupled look-back), then emits out of the same
// window.  What makes that work on an 8-XCD part:
//   * state words are self-contained 64-bit values (status in the top bits, a 32-bit value below), written and polled with
//     RELAXED agent-scope atomics -- no release / acquire fence anywhere, so no L2 write-back per tile (the fences of round
//     1's attempt wrote the XCD's dirty output lines back at every tile: 4-7 ms);
//   * tiles are handed out by a ticket per chunk in the order workgroups START, so the tile a workgroup waits for is
//     always resident or finished -- no assumption about dispatch order or placement;
//   * workgroup b works on chunk b % k: with the usual k = 8 chunks a chunk's tiles (and their state words, and the
//     partial lines of its output buffers) stay on one XCD.
// The arena is laid out from per-column CAPACITIES (the schema's history, KParams::caps) before the launch; a tile that
// would leave its column's capacity raises LF_CAPACITY and emits nothing, and the host repeats the call on the two-pass
// path -- as it does for a schema's first call, for the generic kernels and for offsets / indices beyond 32 bits.
// --------------------------------------------------------------------------
constexpr unsigned long long kLbAgg = 1ull << 62, kLbInc = 2ull << 62;

template <class S>
__device__ __forceinline__ void spec_fused(const KParams& P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (reinterpret_cast<const uint32_t*>(P.first_bad)[2] != 0) return;      // the capacity layout was refused: nothing to emit
  const SpecSmem<S> s(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  RH_MARK_INIT;
  constexpr int T = TileOf<S>::T, NW = TileOf<S>::NW, KP = SCtx<S>::KP;
#ifndef RH_V_NOPRIO
  // Everything up to the look-back is on the critical path of every LATER tile of the chunk (they wait for this tile's
  // totals): those phases run at raised wave priority, the emit walk (nobody waits for it) at the default.
  __builtin_amdgcn_s_setprio(3);
#endif
  // ---- which tile: chunk = b % k, tile of the chunk = the chunk's next ticket
  const uint32_t chunk = blockIdx.x % P.k;
  const uint64_t rows_c = chunk == P.k - 1 ? P.rows_last : P.sz;
  const uint32_t tiles_c = (uint32_t)((rows_c + T - 1) / T);
  if (blockIdx.x / P.k >= tiles_c) return;                                  // (the grid is k x the longest chunk)
  if (tid == 0) {
    s.misc[3] = __hip_atomic_fetch_add(&P.tickets[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.misc[0] = 0xFFFFFFFFu; s.misc[1] = 0; s.misc[2] = 0;
  }
  for (int i = tid; i < S::NNODES; i += T) s.nullcnt[i] = 0;
  for (int i = tid; i < S::NNODES * NW; i += T) s.nullw[i] = 0;
  for (int i = tid; i < S::NBM * kBmWords; i += T) s.bm[i] = 0;
  __syncthreads();
  RH_MARK(0);
  const uint32_t lb = s.misc[3];
  const uint32_t tile = chunk * P.bpc + lb;
  Geo g;
  g.chunk = chunk; g.lrow0 = lb * T; g.rec0 = (uint64_t)chunk * P.sz + g.lrow0;
  { const uint64_t left = rows_c - g.lrow0; g.nrec = left < (uint64_t)T ? (uint32_t)left : (uint32_t)T; }
  uint64_t o0 = 0, o1 = 0;
  if (tid < g.nrec) { o0 = P.offsets[g.rec0 + tid]; o1 = P.offsets[g.rec0 + tid + 1]; }
  const uint64_t wb = P.offsets[g.rec0], we = P.offsets[g.rec0 + g.nrec];
  const uint64_t wb16 = wb & ~15ull;
  const bool fits = (we - wb16) <= (uint64_t)P.win_bytes;
  if (fits) stage_window<T>(P, s.win, wb16, we, tid);
  __syncthreads();
  RH_MARK(1);

  // ---- size walk (the size pass's: fast form, a wave with an anomaly walks again carefully)
  Lane L;
  SCtx<S> c;
  spec_ctx_init(c, P, s, g, tid);
  c.bufp = (const __attribute__((address_space(4))) uint64_t*)(reinterpret_cast<uintptr_t>(P.bufptr) + (size_t)chunk * S::NBUF * 8);
  lane_init_from(L, g, o0, o1, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;
  bool careful = !fits || P.all_careful != 0;
  if constexpr (S::K > 0) {
    spec_run_walk<S, false, false>(P, c, s.win, L, fits, wb16);
    L.redo = L.redo || L.cur > L.end;
    if (__any(L.redo)) {
      careful = true;
      spec_ctx_init(c, P, s, g, tid);
      lane_init_from(L, g, o0, o1, wb16, tid);
      if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;
      spec_run_walk<S, false, true>(P, c, s.win, L, fits, wb16);
    }
    RH_MARK(2);
    if (lane == 0 && careful) atomicOr(&s.misc[2], 2u);       // one wave walked carefully: the whole tile emits carefully
    // ---- wave scan of the K counters (two per dword when they all fit 10 bits: see spec_emit)
    uint32_t allor = 0;
    static_for<0, S::K>([&](auto ik) { allor |= c.cnt[decltype(ik)::value]; });
    if (!__any(allor > 1023u)) {
      constexpr int NDW = (S::K + 1) / 2;
      uint32_t tot[KP] = {};
      static_for<0, NDW>([&](auto id) {
        constexpr int d = decltype(id)::value;
        uint32_t v = c.cnt[2 * d];
        if constexpr (2 * d + 1 < S::K) v |= c.cnt[2 * d + 1] << 16;
        const uint32_t incl = wave_incl_scan(v, lane);
        const uint32_t ex = incl - v;
        c.cnt[2 * d] = ex & 0xFFFFu;
        tot[2 * d] = incl & 0xFFFFu;
        if constexpr (2 * d + 1 < S::K) { c.cnt[2 * d + 1] = ex >> 16; tot[2 * d + 1] = incl >> 16; }
      });
      if (lane == 63) {
        static_for<0, KP / 4>([&](auto iq) {
          constexpr int q = decltype(iq)::value;
          v4w x; x.x = tot[4 * q]; x.y = tot[4 * q + 1]; x.z = tot[4 * q + 2]; x.w = tot[4 * q + 3];
          *reinterpret_cast<v4w*>(s.wtot + wave * KP + 4 * q) = x;
        });
      }
    } else {
      static_for<0, S::K>([&](auto ik) {
        constexpr int k = decltype(ik)::value;
        const uint32_t v = c.cnt[k];
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s.wtot[wave * KP + k] = incl;
        c.cnt[k] = incl - v;
      });
    }
  }
  // errors of the size walk (malformed records): reported like the size pass does; the tile still takes part in the scan
  if (L.err) atomicMin(&s.misc[0], tid);
  RH_MARK(3);
  __syncthreads();                                            // wtot, misc[0], misc[2]
  RH_MARK(4);
  if (s.misc[0] == tid) {
    ErrInfo ei; ei.code = L.err; ei.pad = 0; ei.detail = L.edetail;
    P.errinfo[tile] = ei;
    atomicMax(P.first_bad, ~(unsigned long long)(g.rec0 + tid));
  }
  careful = careful || (s.misc[2] & 2u) != 0 || s.misc[0] != 0xFFFFFFFFu;

  // ---- look-back (wave 0): this tile's totals out, the chunk-relative prefix of the tiles in front in.  Lane (j, k) polls
  //      counter k's word of the (back + j)-th tile in front, NJ = 64 / K tiles per step: the tiles right in front are usually
  //      still at their totals and an inclusive prefix sits a few tiles back, so a step -- ONE round trip to the memory side
  //      -- normally ends the search, where walking back tile by tile paid a round trip per tile (~9 k cycles per tile,
  //      profiles/r04p_*).  Lane k then adds the step's words up to the first inclusive one.
  if constexpr (S::K > 0) {
    if (wave == 0) {
      static_assert(S::K <= 64, "the look-back wave holds one lane per counter");
      constexpr uint32_t NJ = 64 / S::K;
      const uint32_t k_of = lane % (uint32_t)S::K, j_of = lane / (uint32_t)S::K;
      const bool in_grid = j_of < NJ;
      uint32_t agg = 0;
      if (lane < (uint32_t)S::K)
        for (int w = 0; w < NW; w++) agg += s.wtot[w * KP + lane];
      unsigned long long* const mine = P.lookback + (size_t)tile * S::K + k_of;
      if (lane < (uint32_t)S::K)
        __hip_atomic_store(mine, (lb == 0 ? kLbInc : kLbAgg) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t excl = 0;
      bool found = lb == 0, stuck = false;
      uint32_t* const xw = s.gbx + KP;                        // [64][2] exchange area of the look-back wave
      for (uint32_t back = 1;; back += NJ) {
        if (!__any(lane < (uint32_t)S::K && !found)) break;
        uint32_t val = 0, st = 2;                             // in front of the chunk's first tile: an inclusive 0
        const uint32_t dist = back + j_of;
        if (in_grid && dist <= lb) {
          const unsigned long long* q = mine - (size_t)dist * S::K;
          unsigned long long w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (the tile polled holds an earlier ticket of this chunk: its workgroup is running or done, so the word WILL come.
          //  The bound -- seconds -- only keeps a fault elsewhere from turning into a hung GPU: the call then fails over to
          //  the two-pass path through LF_CAPACITY.)
          for (uint32_t spins = 0; (w >> 62) == 0 && spins < (1u << 24); spins++) {
            __builtin_amdgcn_s_sleep(1);
            w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((w >> 62) == 0) { stuck = true; w = kLbInc; }
          val = (uint32_t)w; st = (uint32_t)(w >> 62);
        }
        if (in_grid) { xw[2 * lane] = val; xw[2 * lane + 1] = st; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (a wave's DS instructions execute in order)
        if (lane < (uint32_t)S::K && !found) {
          for (uint32_t j = 0; j < NJ && !found; j++) {
            excl += xw[2 * (j * S::K + lane)];
            found = xw[2 * (j * S::K + lane) + 1] == 2u;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      uint32_t over = 0;
      if (lane < (uint32_t)S::K) {
        if (lb != 0) __hip_atomic_store(mine, kLbInc | (unsigned long long)(uint32_t)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t end = (uint64_t)excl + agg;
        over = end > P.caps[(size_t)lane * P.k + chunk] ? 1u : 0u;
        if (lb == tiles_c - 1) P.totals[(size_t)lane * P.k + chunk] = end;
        s.gbx[lane] = excl;                                   // chunk-relative base of this tile, per counter
      }
      if (__any(over != 0 || stuck)) {
        if (lane == 0) {
          s.misc[1] = 1;
          atomicOr(reinterpret_cast<uint32_t*>(P.first_bad) + 2, (uint32_t)LF_CAPACITY);
        }
      }
    }
    RH_MARK(5);
    __syncthreads();
    RH_MARK(6);
    if (s.misc[1] != 0) return;                               // over capacity: the host repeats the call on the two-pass path
    static_for<0, S::K>([&](auto ik) {
      constexpr int k = decltype(ik)::value;
      c.gb[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.gbx[k]);
    });
    if constexpr (NW > 1) {
      // (v_readlane selects among 64 lanes: schemas with more counters -- up to kMaxCounters = 96 -- take them 64 at a time)
      static_for<0, (S::K + 63) / 64>([&](auto ig) {
        constexpr int k0 = decltype(ig)::value * 64;
        constexpr int kn = S::K - k0 < 64 ? S::K - k0 : 64;
        uint32_t prev = 0;
        if (lane < (uint32_t)kn) {
          for (int w = 0; w < NW - 1; w++) prev += (int)wave > w ? s.wtot[w * KP + k0 + lane] : 0u;
        }
        static_for<0, kn>([&](auto ik) {
          constexpr int k = decltype(ik)::value;
          c.cnt[k0 + k] += (uint32_t)__builtin_amdgcn_readlane((int)prev, k);
        });
      });
    }
  }

  // ---- emit walk out of the same window
#ifndef RH_V_NOPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  RH_MARK(7);
  lane_init_from(L, g, o0, o1, wb16, tid);
  if (L.live && (we - wb16) > 0xFFFFFFF0ull) L.err = E_EOB;
  if (careful) {
    spec_run_walk<S, true, true>(P, c, s.win, L, fits, wb16);
  } else {
    spec_run_walk<S, true, false>(P, c, s.win, L, fits, wb16);
    if (L.redo) L.err = E_INTERNAL;
  }
  RH_MARK(8);
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  __syncthreads();
  report_errors(P, s.misc, L, g, tid, tile);
  for (int i = tid; i < S::NNODES; i += T) {
    uint32_t v = s.nullcnt[i];
    for (int w = 0; w < NW; w++) v += s.nullw[i * NW + w];
    if (v) atomicAdd(&P.nullcount[((size_t)i * P.k + g.chunk) * P.null_slots + (tile & (P.null_slots - 1))], v);
  }
  if constexpr (S::NB0 > 0) {
    for (uint32_t i = tid; i < (uint32_t)(S::NB0 * NW); i += T) {
      const uint32_t slot = i / NW, w = i % NW;
      if (w * 64u < g.nrec) st_global<uint64_t, false>(c.buf(S::bm0buf(slot)), (g.lrow0 >> 6) + w, c.bmw0[i]);
    }
  }
  if constexpr (S::NBM > 0) {
    static_for<0, S::NBM>([&](auto ib) {
      constexpr int slot = decltype(ib)::value;
      const uint32_t w0 = c.gb[S::bmdom(slot) - 1] >> 5;
      for (int w = tid; w < kBmWords; w += T) {
        const uint32_t v = s.bm[slot * kBmWords + w];
        if (v) atomic_or_global(c.buf(S::bmbuf(slot)), (uint64_t)w0 + w, v);
      }
    });
  }
  RH_MARK(9);
  RH_MARK_FLUSH;
}

}  // namespace rh
