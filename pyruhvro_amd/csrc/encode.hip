// CDNA4 (gfx950) kernels of the Arrow -> Avro encode path, generic form: the schema
// program is INTERPRETED wave-uniformly (one scalar Op fetch per step), row cursors live in LDS.  Works for every
// schema without a compile step; the schema-specialised form (specialize.cpp -> hiprtc) runs the program unrolled
// with the loads of many fields in flight at once.  Both share the byte sinks and the kernel frame (encode_walk.h).
//
//   rh_e_size   walk 1: encoded length of every row -> rowlen[], per-workgroup sums (+ first failing row)
//   rh_k_scan   (kernels.hip) chunk-segmented exclusive scan of those sums
//   rh_e_emit   scan inside the workgroup, offsets[row+1], walk 2 stages the bytes in LDS, coalesced copy-out
//
// Byte shuffling bound by HBM and dependent-load latency; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "encode_walk.h"

namespace rh {

typedef const __attribute__((address_space(4))) Op* ProgPtr;
__device__ __forceinline__ Op ld_op(ProgPtr p) {      // (field by field: see kernels.hip)
  Op o;
  o.code = p->code; o.flags = p->flags; o.dom = p->dom; o.a = p->a; o.b = p->b; o.c = p->c;
  o.buf0 = p->buf0; o.buf1 = p->buf1; o.buf2 = p->buf2; o.node = p->node;
  return o;
}


struct ECtx {
  const EParams* P;
  uint32_t* idx;        // LDS [ndom][256] current row of this lane in every row domain
  uint32_t* remv;       // LDS [depth][256] items left in the current list
  RH_GLOBAL uint8_t* out;   // emit, direct form: this row's first output byte in HBM
  RH_LDS uint8_t* lout;     // emit, staged form: this row's first output byte in the LDS window
  RH_LDS uint8_t* stage;    // unused by the interpreter (the specialised kernels' string staging area)
  uint32_t tid;
  __device__ __forceinline__ uint32_t& row(int dom) const { return idx[dom * kBlock + tid]; }
  __device__ __forceinline__ uint32_t& remaining(int d) const { return remv[d * kBlock + tid]; }
  __device__ __forceinline__ uint64_t in(int buf) const { return P->in_ptr[buf]; }
  __device__ __forceinline__ bool bit(int buf, uint32_t r) const {   // validity / boolean value of logical row r
    const uint32_t b = r + P->in_bitoff[buf];
    return (reinterpret_cast<const RH_GLOBAL uint8_t*>(P->in_ptr[buf])[b >> 3] >> (b & 7)) & 1;
  }
};

// The interpreter reads a node's inputs only on the lanes that write it (one Op at a time, nothing to overlap),
// and shares the byte sinks (put_*) with the specialised kernels.
struct EInterp {
  using Ctx = ECtx;
  static __device__ __forceinline__ uint32_t cursor_words(const EParams& P) {
    return (uint32_t)((P.ndom > 0 ? P.ndom : 1) + (P.list_depth > 0 ? P.list_depth : 1)) * kBlock;
  }
  static __device__ __forceinline__ void init(ECtx& c, const EParams& P, uint32_t* cursors, const Geo& g, uint32_t tid) {
    const int nd = P.ndom > 0 ? P.ndom : 1;
    c.P = &P; c.idx = cursors; c.remv = cursors + nd * kBlock;
    c.out = nullptr; c.lout = nullptr; c.tid = tid;
    c.idx[tid] = (uint32_t)(g.rec0 + tid);        // domain 0: the row of the batch
  }
  template <int MODE>
  static __device__ __forceinline__ void walk(ECtx& c, ELane& L) {
    const EParams& P = *c.P;
    // (the program through the constant address space, the next op requested while this one runs: kernels.hip `walk`)
    const ProgPtr prog = reinterpret_cast<ProgPtr>(reinterpret_cast<uintptr_t>(P.prog));
    int pc = 0;
    Op nxt = ld_op(prog);
    for (;;) {
      pc = __builtin_amdgcn_readfirstlane(pc);
      const Op op = nxt;
      asm volatile("" ::"s"(op.code), "s"(op.flags), "s"(op.dom), "s"(op.a), "s"(op.b), "s"(op.c), "s"(op.buf0), "s"(op.buf1), "s"(op.buf2), "s"(op.node));
      if (op.code == OP_END) return;
      int npc = op.code == OP_LIST_TAIL ? op.b : pc + 1;
      nxt = ld_op(prog + npc);
      const bool wr = L.writes();          // this lane writes this node's bytes
      switch (op.code) {

        case OP_FIXED: {                      // fast_encode.rs:391-399, 407-455
          if (wr) {
            const uint32_t r = c.row(op.dom);
            bool isnull = false;
            if (op.flags & F_NULLABLE) {
              isnull = !c.bit(op.buf0, r);
              put_branch<MODE>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
            }
            if (!isnull) {
              if (op.a == FK_I32) put_varint<MODE>(c, L, (int64_t)reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1))[r]);
              else if (op.a == FK_I64) put_varint<MODE>(c, L, reinterpret_cast<const RH_GLOBAL int64_t*>(c.in(op.buf1))[r]);
              else if (op.a == FK_F32) put_raw<MODE, 4>(c, L, reinterpret_cast<const RH_GLOBAL uint32_t*>(c.in(op.buf1))[r]);
              else if (op.a == FK_F64) put_raw<MODE, 8>(c, L, reinterpret_cast<const RH_GLOBAL uint64_t*>(c.in(op.buf1))[r]);
              else put_byte<MODE>(c, L, c.bit(op.buf1, r) ? 1 : 0);
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
              put_branch<MODE>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
            }
            if (!isnull) {
              const RH_GLOBAL int32_t* off = reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1));
              const uint32_t s0 = (uint32_t)off[r], s1 = (uint32_t)off[r + 1];
              const RH_GLOBAL uint8_t* sp = reinterpret_cast<const RH_GLOBAL uint8_t*>(c.in(op.buf2)) + s0;
              const uint32_t n = s1 - s0;
              if (op.code == OP_STRING) {
                put_varint<MODE>(c, L, (int64_t)n);
                put_bytes<MODE>(c, L, sp, n);
              } else {
                const int32_t found = EnumTableFinder{P.sym_off, P.sym_data, op.b, op.c}(sp, n);
                if (found < 0) { L.err = EE_ENUM; L.eop = (uint32_t)pc; L.edetail = r; }
                else put_varint<MODE>(c, L, found);
              }
            }
          }
          break;
        }

        case OP_BIN: {                        // fixed / decimal / uuid / duration (encode_walk.h e_bin_put)
          if (wr) {
            BinV v;
            v.lo = 0; v.hi = 0;
            const uint32_t r = c.row(op.dom);
            v.valid = (op.flags & F_NULLABLE) ? c.bit(op.buf0, r) : true;
            if (op.a == BN_DURATION && v.valid) {             // Duration(ms): one i64 per row
              v.lo = reinterpret_cast<const RH_GLOBAL u64u*>(c.in(op.buf1))[r];
            } else if (op.a != BN_FIXED && v.valid) {
              const RH_GLOBAL u64u* p = reinterpret_cast<const RH_GLOBAL u64u*>(c.in(op.buf1)) + 2ull * r;
              v.lo = p[0]; v.hi = p[1];
            }
            e_bin_put<MODE>(c, L, op, v);
          }
          break;
        }

        case OP_REC_BEGIN: {                  // NullableRecord, 465-476: the struct's own validity
          L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
          bool present = false;
          if (wr) {
            const bool isnull = !c.bit(op.buf0, c.row(op.dom));
            put_branch<MODE>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
            present = !isnull;
          }
          L.pres = present;
          break;
        }
        case OP_REC_END: e_rec_end(L); break;

        case OP_UNION_BEGIN: {                // UnionEncoder::write, 507-521 (sparse: children share the row)
          L.pstk = (L.pstk << 1) | (L.pres ? 1u : 0u);
          L.sstk = (L.sstk << 8) | 0xFFull;
          if (wr) {
            const int32_t t = reinterpret_cast<const RH_GLOBAL int8_t*>(c.in(op.buf1))[c.row(op.dom)];
            if (t < 0 || t >= op.a) { L.err = EE_UNION; L.eop = (uint32_t)pc; L.edetail = t; }
            else {
              put_varint<MODE>(c, L, t);
              L.sstk = (L.sstk & ~0xFFull) | (uint64_t)t;
            }
          }
          break;
        }
        case OP_VARIANT: e_variant(L, op); break;
        case OP_UNION_END: e_union_end(L); break;

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
              put_branch<MODE>(c, L, isnull, (op.flags & F_NULL_FIRST) != 0);
            }
            if (!isnull) {
              const RH_GLOBAL int32_t* off = reinterpret_cast<const RH_GLOBAL int32_t*>(c.in(op.buf1));
              const uint32_t s0 = (uint32_t)off[r], s1 = (uint32_t)off[r + 1];
              n = s1 - s0;
              if (n > 0) put_varint<MODE>(c, L, (int64_t)n);
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
          if (!__any(item)) { npc = op.b; nxt = ld_op(prog + npc); break; }
          L.pres = item;
          break;
        }
        case OP_LIST_TAIL: {
          if (L.live && L.err == 0 && c.remaining(op.c) > 0) {
            c.remaining(op.c) -= 1;
            c.row(op.a) += 1;
          }
          break;                               // (npc = op.b: back to LIST_NEXT)
        }
        case OP_LIST_END: e_list_end<MODE>(c, L); break;
        default: return;
      }
      pc = npc;
    }
  }
};

extern "C" __global__ void __launch_bounds__(kBlock) rh_e_size(EParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  e_size_body<EInterp>(P, smem);
}
extern "C" __global__ void __launch_bounds__(kBlock) rh_e_emit(EParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  e_emit_body<EInterp>(P, smem);
}

}  // namespace rh

// LDS bytes in front of the staging window for the generic kernels
extern "C" uint32_t rh_enc_lds_bytes(int ndom, int list_depth) {
  return rh::enc_lds_fixed_bytes((uint32_t)((ndom > 0 ? ndom : 1) + (list_depth > 0 ? list_depth : 1)) * rh::kBlock);
}
extern "C" int rh_launch_esize(const rh::EParams* P, uint32_t lds_bytes, void* stream) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipLaunchKernelGGL(rh::rh_e_size, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
extern "C" int rh_launch_eemit(const rh::EParams* P, uint32_t lds_bytes, void* stream) {
  (void)hipGetLastError();      // (an earlier, unrelated error -- a hipStreamQuery that said "not ready" -- must not be read as this launch\'s)
  hipLaunchKernelGGL(rh::rh_e_emit, dim3(P->nblocks), dim3(rh::kBlock), lds_bytes, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
