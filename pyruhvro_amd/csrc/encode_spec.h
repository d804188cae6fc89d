// Kernel frame of the schema-specialised Arrow -> Avro kernels.  specialize.cpp generates, per schema, a struct
//   struct ESpec { static constexpr int NDOM, DEPTH;  template <int MODE, class Ctx> static void walk(Ctx&, ELane&); };
// whose walk() is the schema program unrolled over the handlers of encode_walk.h with constexpr Ops: the loads of
// every field of a row domain are issued together (e_*_load) before the first e_*_put runs, and the row cursors
// live in registers.  hiprtc compiles it for gfx950; the code object is cached like the decode kernels'.
#pragma once
#include "encode_walk.h"

namespace rh {

template <int I>
struct EIC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void e_static_for(F&& f) {
  if constexpr (I < N) {
    f(EIC<I>{});
    e_static_for<I + 1, N>(f);
  }
}

// the input tables are read-only for the whole launch: constant address space -> scalar loads the compiler may
// hoist and merge across the walk's stores
typedef const __attribute__((address_space(4))) uint64_t* ecp64;
typedef const __attribute__((address_space(4))) uint32_t* ecp32;

template <class S>
struct ESCtx {
  static constexpr int ND = S::NDOM > 0 ? S::NDOM : 1, DP = S::DEPTH > 0 ? S::DEPTH : 1;
  uint32_t rowv[ND];    // current row of this lane in every row domain (registers: only constant indices below)
  uint32_t remv[DP];    // items left in the current list, per nesting level
  ecp64 in_ptr;
  ecp32 in_bitoff;
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  RH_GLOBAL uint8_t* out;
  RH_LDS uint8_t* lout;
  RH_LDS uint8_t* stage;    // this wave's string staging area (e_string_cofetch), null = none
  __device__ __forceinline__ uint32_t row(int dom) const {
    uint32_t r = 0;
    e_static_for<0, ND>([&](auto i) { if (decltype(i)::value == dom) r = rowv[decltype(i)::value]; });
    return r;
  }
  __device__ __forceinline__ void set_row(int dom, uint32_t v) {
    e_static_for<0, ND>([&](auto i) { if (decltype(i)::value == dom) rowv[decltype(i)::value] = v; });
  }
  __device__ __forceinline__ uint32_t rem(int d) const {
    uint32_t r = 0;
    e_static_for<0, DP>([&](auto i) { if (decltype(i)::value == d) r = remv[decltype(i)::value]; });
    return r;
  }
  __device__ __forceinline__ void set_rem(int d, uint32_t v) {
    e_static_for<0, DP>([&](auto i) { if (decltype(i)::value == d) remv[decltype(i)::value] = v; });
  }
  __device__ __forceinline__ uint64_t in(int buf) const { return in_ptr[buf]; }
  __device__ __forceinline__ uint32_t bitoff(int buf) const { return in_bitoff[buf]; }
};

template <class S>
struct ESpecW {
  using Ctx = ESCtx<S>;
  static __device__ __forceinline__ uint32_t cursor_words(const EParams&) { return 0; }
  static __device__ __forceinline__ void init(Ctx& c, const EParams& P, uint32_t*, const Geo& g, uint32_t tid) {
    e_static_for<0, Ctx::ND>([&](auto i) { c.rowv[decltype(i)::value] = 0; });
    e_static_for<0, Ctx::DP>([&](auto i) { c.remv[decltype(i)::value] = 0; });
    const uint64_t r0 = g.rec0 + tid;
    c.rowv[0] = (uint32_t)(r0 < P.n ? r0 : P.n - 1);          // every lane reads a valid row (encode_walk.h)
    c.in_ptr = (ecp64)(uintptr_t)P.in_ptr;
    c.in_bitoff = (ecp32)(uintptr_t)P.in_bitoff;
    c.sym_off = P.sym_off; c.sym_data = P.sym_data; c.out = nullptr; c.lout = nullptr; c.stage = nullptr;
  }
  template <int MODE>
  static __device__ __forceinline__ void walk(Ctx& c, ELane& L) { S::template walk<MODE>(c, L); }
};

}  // namespace rh
