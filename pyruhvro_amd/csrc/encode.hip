// CDNA4 (gfx950) kernels of the Arrow -> Avro encode path: the mirror of the
// direct decode, behind pyruhvro's serialize_record_batch.
//
// Reference: ruhvro/src/fast_encode.rs:387-599 (per-row write of every encoder variant, zig-zag varints,
// one block per array/map) and ruhvro/src/serialize.rs:19-67 (chunking).  One lane = one row of the batch,
// one workgroup = 256 consecutive rows of one output chunk; the SAME schema program as the decoder
// (program.h) is interpreted wave-uniformly, its buffer ids now naming the INPUT Arrow buffers.  The host
// hands every buffer over rebased to logical row 0 (engine_encode.cpp), so rows index buffers directly.
//
//   rh_e_size   walk 1: encoded length of every row -> per-workgroup sums (+ first failing row)
//   rh_k_scan   (kernels.hip) chunk-segmented exclusive scan of those sums
//   rh_e_emit   walk 1 again, scan inside the workgroup, walk 2 writes offsets[row+1] and the datum bytes
//
// Byte shuffling bound by HBM and per-lane store issue; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "encode.h"
#include "kernel_common.h"

namespace rh {

struct ELane {
  uint32_t len;        // bytes written so far by this row (size pass: counted; emit pass: cursor)
  uint32_t err;
  int64_t edetail;
  uint32_t eop;
  bool live, pres;
  uint32_t pstk, lstk;
  uint64_t sstk;
};

struct ECtx {
  const EParams* P;
  uint32_t* idx;        // LDS [ndom][256] current row of this lane in every row domain
  uint32_t* rem;        // LDS [depth][256] items left in the current list
  RH_GLOBAL uint8_t* out;   // emit: this row's first output byte (nullptr in the size pass)
  uint32_t tid;
  __device__ __forceinline__ uint32_t& row(int dom) const { return idx[dom * kBlock + tid]; }
  __device__ __forceinline__ uint32_t& remaining(int d) const { return rem[d * kBlock + tid]; }
  __device__ __forceinline__ uint64_t in(int buf) const { return P->in_ptr[buf]; }
  __device__ __forceinline__ bool bit(int buf, uint32_t r) const {   // validity / boolean value of logical row r; absent buffer = all set
    const uint64_t p = P->in_ptr[buf];
    if (!p) return true;
    const uint32_t b = r + P->in_bitoff[buf];
    return (reinterpret_cast<const RH_GLOBAL uint8_t*>(p)[b >> 3] >> (b & 7)) & 1;
  }
};

// write_zigzag_long, fast_encode.rs:583-591
template <bool EMIT>
__device__ __forceinline__ void put_varint(const ECtx& c, ELane& L, int64_t v) {
  uint64_t zz = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  for (;;) {
    const bool more = (zz & ~0x7Full) != 0;
    if (EMIT) c.out[L.len] = (uint8_t)((zz & 0x7F) | (more ? 0x80 : 0));
    L.len++;
    if (!more) break;
    zz >>= 7;
  }
}

template <bool EMIT>
__device__ __forceinline__ void put_bytes(const ECtx& c, ELane& L, const RH_GLOBAL uint8_t* s, uint32_t n) {
  if (EMIT) {
    RH_GLOBAL uint8_t* d = c.out + L.len;
    uint32_t j = 0;
    for (; j + 8 <= n; j += 8) *reinterpret_cast<RH_GLOBAL u64u*>(d + j) = *reinterpret_cast<const RH_GLOBAL u64u*>(s + j);
    for (; j < n; j++) d[j] = s[j];
  }
  L.len += n;
}

// write_nullable (563-572): branch index for a null / a value
template <bool EMIT>
__device__ __forceinline__ void put_branch(const ECtx& c, ELane& L, bool is_null, bool null_first) {
  put_varint<EMIT>(c, L, is_null ? (null_first ? 0 : 1) : (null_first ? 1 : 0));
}

template <bool EMIT>
__device__ __forceinline__ void ewalk(const ECtx& c, ELane& L) {
  const EParams& P = *c.P;
  int pc = 0;
  for (;;) {
    pc = __builtin_amdgcn_readfirstlane(pc);
    const Op op = P.prog[pc];
    const bool act = L.live && L.err == 0;
    const bool wr = act && L.pres;          // this lane writes this node's bytes
    switch (op.code) {
      case OP_END: return;

      case OP_FIXED: {                      // fast_encode.rs:391-399, 407-455
        if (wr) {
          const uint32_t r = c.row(op.dom);
          bool isnull = false;
          if (op.flags & F_NULLABLE) {
            isnull = !c.bit(op.buf0, r);
            put_branch<EMIT>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
          }
          if (!isnull) {
            if (op.a == FK_I32) put_varint<EMIT>(c, L, (int64_t)reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1))[r]);
            else if (op.a == FK_I64) put_varint<EMIT>(c, L, reinterpret_cast<const RH_GLOBAL int64_t*>(c.in(op.buf1))[r]);
            else if (op.a == FK_F32) put_bytes<EMIT>(c, L, reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf1)) + (size_t)r * 4, 4);
            else if (op.a == FK_F64) put_bytes<EMIT>(c, L, reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf1)) + (size_t)r * 8, 8);
            else {
              if (EMIT) c.out[L.len] = c.bit(op.buf1, r) ? 1 : 0;
              L.len++;
            }
          }
        }
        break;
      }

      case OP_STRING:
      case OP_ENUM: {                       // write_string 593-597, write_enum_idx 574-581
        if (wr) {
          const uint32_t r = c.row(op.dom);
          bool isnull = false;
          if (op.flags & F_NULLABLE) {
            isnull = !c.bit(op.buf0, r);
            put_branch<EMIT>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
          }
          if (!isnull) {
            const RH_GLOBAL int32_t* off = reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1));
            const uint32_t s0 = (uint32_t)off[r], s1 = (uint32_t)off[r + 1];
            const RH_GLOBAL uint8_t* sp = reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + s0;
            const uint32_t n = s1 - s0;
            if (op.code == OP_STRING) {
              put_varint<EMIT>(c, L, (int64_t)n);
              put_bytes<EMIT>(c, L, sp, n);
            } else {
              int32_t found = -1;
              for (int32_t s = 0; s < op.c && found < 0; s++) {
                const uint32_t a = P.sym_off[op.b + s], b = P.sym_off[op.b + s + 1];
                if (b - a != n) continue;
                bool eq = true;
                for (uint32_t j = 0; j < n && eq; j++) eq = sp[j] == P.sym_data[a + j];
                if (eq) found = s;
              }
              if (found < 0) { L.err = EE_ENUM; L.eop = (uint32_t)pc; L.edetail = r; }
              else put_varint<EMIT>(c, L, found);
            }
          }
        }
        break;
      }

      case OP_REC_BEGIN: {                  // NullableRecord, 465-476: the struct's own validity
        L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
        bool present = false;
        if (wr) {
          const bool isnull = !c.bit(op.buf0, c.row(op.dom));
          put_branch<EMIT>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
          present = !isnull;
        }
        L.pres = present;
        break;
      }
      case OP_REC_END:
        L.pres = L.pstk & 1;
        L.pstk >>= 1;
        break;

      case OP_UNION_BEGIN: {                // UnionEncoder::write, 507-521 (sparse: children share the row)
        L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
        L.sstk = (L.sstk << 8) | 0xFFull;
        if (wr) {
          const int32_t t = reinterpret_cast<const RH_GLOBAL int8_t*>(c.in(op.buf1))[c.row(op.dom)];
          if (t < 0 || t >= op.a) { L.err = EE_UNION; L.eop = (uint32_t)pc; L.edetail = t; }
          else {
            put_varint<EMIT>(c, L, t);
            L.sstk = (L.sstk & ~0xFFull) | (uint64_t)t;
          }
        }
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

      case OP_LIST_BEGIN: {                 // ListEncoder / MapEncoder::write, 525-561 (+ Nullable*, 478-496)
        L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
        L.lstk = (L.lstk << 1) | (L.live ? 1u : 0u);
        uint32_t n = 0;
        bool has = false;
        if (wr) {
          const uint32_t r = c.row(op.dom);
          bool isnull = false;
          if (op.flags & F_NULLABLE) {
            isnull = !c.bit(op.buf0, r);
            put_branch<EMIT>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
          }
          if (!isnull) {
            const RH_GLOBAL int32_t* off = reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1));
            const uint32_t s0 = (uint32_t)off[r], s1 = (uint32_t)off[r + 1];
            n = s1 - s0;
            if (n > 0) put_varint<EMIT>(c, L, (int64_t)n);
            c.row(op.a) = s0;               // op.a = child row domain: first item
            has = true;
          }
        }
        c.remaining(op.c) = n;
        // lstk bit 0 remembers "this lane owes a 0 terminator"; the saved `live` sits one bit above it
        L.lstk = (L.lstk << 1) | (has ? 1u : 0u);
        L.live = has;
        L.pres = has;
        break;
      }
      case OP_LIST_NEXT: {
        const bool item = L.live && L.err == 0 && c.remaining(op.c) > 0;
        if (!__any(item)) { pc = op.b; continue; }
        L.pres = item;
        break;
      }
      case OP_LIST_TAIL: {
        if (L.live && L.err == 0 && c.remaining(op.c) > 0) {
          c.remaining(op.c) -= 1;
          c.row(op.a) += 1;
        }
        pc = op.b;
        continue;
      }
      case OP_LIST_END: {
        const bool owes = (L.lstk & 1) != 0;
        L.lstk >>= 1;
        L.live = L.lstk & 1;
        L.lstk >>= 1;
        L.pres = L.pstk & 1;
        L.pstk >>= 1;
        if (owes && L.live && L.err == 0) put_varint<EMIT>(c, L, 0);    // terminator (an empty list is just this 0)
        break;
      }
      default: return;
    }
    pc++;
  }
}

__host__ __device__ inline uint32_t enc_lds_words(int ndom, int list_depth) {
  return (uint32_t)(ndom > 0 ? ndom : 1) * kBlock + (uint32_t)(list_depth > 0 ? list_depth : 1) * kBlock + 8;
}
extern "C" uint32_t rh_enc_lds_bytes(int ndom, int list_depth) { return enc_lds_words(ndom, list_depth) * 4; }

struct ESmem {
  uint32_t* idx;
  uint32_t* rem;
  uint32_t* misc;   // [0] lowest erroring tid, [4..7] wave totals
};
__device__ __forceinline__ ESmem ecarve(const EParams& P, uint8_t* smem) {
  ESmem s;
  uint32_t* p = reinterpret_cast<uint32_t*>(smem);
  s.idx = p; p += (P.ndom > 0 ? P.ndom : 1) * kBlock;
  s.rem = p; p += (P.list_depth > 0 ? P.list_depth : 1) * kBlock;
  s.misc = p;
  return s;
}

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

__device__ __forceinline__ void elane_init(ELane& L, const ESmem& s, const EParams& P, const Geo& g, uint32_t tid) {
  L.len = 0; L.err = 0; L.edetail = 0; L.eop = 0;
  L.live = tid < g.nrec; L.pres = L.live;
  L.pstk = 0; L.lstk = 0; L.sstk = 0;
  s.idx[tid] = (uint32_t)(g.rec0 + tid);        // domain 0: the row of the batch
}

__device__ __forceinline__ void ereport(const EParams& P, const ESmem& s, const ELane& L, const Geo& g, uint32_t tid) {
  if (L.err) atomicMin(&s.misc[0], tid);
  __syncthreads();
  if (s.misc[0] == tid) {
    ErrInfo ei; ei.code = L.err; ei.pad = L.eop; ei.detail = L.edetail;
    P.errinfo[blockIdx.x] = ei;
    atomicMax(P.first_bad, ~(unsigned long long)(g.rec0 + tid));
  }
}

extern "C" __global__ void __launch_bounds__(kBlock) rh_e_size(EParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const ESmem s = ecarve(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = egeometry(P, blockIdx.x);
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  ELane L;
  elane_init(L, s, P, g, tid);
  __syncthreads();
  ECtx c{&P, s.idx, s.rem, nullptr, tid};
  ewalk<false>(c, L);
  const uint32_t v = wave_sum(L.live || tid < g.nrec ? L.len : 0u);
  if (lane == 0) s.misc[4 + wave] = v;
  ereport(P, s, L, g, tid);     // barrier inside
  if (tid == 0) P.blocksum[blockIdx.x] = s.misc[4] + s.misc[5] + s.misc[6] + s.misc[7];
}

extern "C" __global__ void __launch_bounds__(kBlock) rh_e_emit(EParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const ESmem s = ecarve(P, smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Geo g = egeometry(P, blockIdx.x);
  if (tid == 0) s.misc[0] = 0xFFFFFFFFu;
  ELane L;
  elane_init(L, s, P, g, tid);
  __syncthreads();
  ECtx c{&P, s.idx, s.rem, nullptr, tid};
  ewalk<false>(c, L);                              // this row's length again (cheaper than an n x 4 B round trip)
  const uint32_t mylen = tid < g.nrec ? L.len : 0u;
  const uint32_t incl = wave_incl_scan(mylen, lane);
  if (lane == 63) s.misc[4 + wave] = incl;
  __syncthreads();
  uint32_t base = P.blockbase[blockIdx.x];
  for (uint32_t w = 0; w < wave; w++) base += s.misc[4 + w];
  const uint32_t start = base + incl - mylen;      // chunk-relative byte offset of this row's datum
  RH_GLOBAL int32_t* offs = reinterpret_cast<RH_GLOBAL int32_t*>(reinterpret_cast<uintptr_t>(P.outptr[(size_t)g.chunk * 2]));
  RH_GLOBAL uint8_t* data = reinterpret_cast<RH_GLOBAL uint8_t*>(reinterpret_cast<uintptr_t>(P.outptr[(size_t)g.chunk * 2 + 1]));
  if (tid < g.nrec) offs[g.lrow0 + tid + 1] = (int32_t)(start + mylen);
  if (g.lrow0 == 0 && tid == 0) offs[0] = 0;
  __syncthreads();                                 // idx / rem are re-initialised below
  elane_init(L, s, P, g, tid);
  __syncthreads();
  c.out = data + start;
  ewalk<true>(c, L);
  ereport(P, s, L, g, tid);
}

}  // namespace rh

extern "C" int rh_launch_esize(const rh::EParams* P, uint32_t lds_bytes, void* stream) {
  hipLaunchKernelGGL(rh::rh_e_size, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_eemit(const rh::EParams* P, uint32_t lds_bytes, void* stream) {
  hipLaunchKernelGGL(rh::rh_e_emit, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
