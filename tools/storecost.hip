// microbench: what one wave-level global STORE instruction costs a CU, by width / pitch / alignment / active lanes,
// with the destination resident in L2 (each wave rewrites a private 4 KiB region) so that HBM is not what is measured.
// Also: VALU issue rate of N waves per SIMD running dependent integer chains (is a SIMD's VALU one-per-quad-cycle?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint16_t __attribute__((aligned(1))) u16u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t v4w __attribute__((ext_vector_type(4)));
typedef v4w __attribute__((aligned(1))) v4wu;
#define GL __attribute__((address_space(1)))

template <int W>
__device__ __forceinline__ void st(uintptr_t base, uint32_t off, uint32_t v) {
  if (W == 1) *reinterpret_cast<GL uint8_t*>(base + off) = (uint8_t)v;
  if (W == 2) *reinterpret_cast<GL u16u*>(base + off) = (uint16_t)v;
  if (W == 4) *reinterpret_cast<GL u32u*>(base + off) = v;
  if (W == 8) *reinterpret_cast<GL u64u*>(base + off) = ((uint64_t)v << 32) | v;
  if (W == 16) { v4w x; x.x = v; x.y = v + 1; x.z = v + 2; x.w = v + 3; *reinterpret_cast<GL v4wu*>(base + off) = x; }
}

// every wave: `iters` stores of W bytes per active lane at region + ((i * 64 * pitch) & (R-1)) + lane * pitch + mis
template <int W>
__global__ void k_store(uint8_t* buf, uint32_t region, uint32_t pitch, uint32_t mis, uint32_t lane_mod, int iters) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const uintptr_t base = reinterpret_cast<uintptr_t>(buf) + (size_t)wave * (region + 4096);
  const bool act = (lane % lane_mod) == 0;
  uint32_t off = lane * pitch + mis;
  for (int i = 0; i < iters; i++) {
    if (act) st<W>(base, off & (region - 1), (uint32_t)i);
    off += 64 * pitch;
  }
}

// two stores per iteration: the second one `delta` bytes after the first (same cache lines when delta is small: the
// head + overlapping-tail pair of a short string copy), or in the other half of the region (different lines)
template <int W>
__global__ void k_store_pair(uint8_t* buf, uint32_t region, uint32_t pitch, uint32_t delta, int iters) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const uintptr_t base = reinterpret_cast<uintptr_t>(buf) + (size_t)wave * (region + 4096);
  uint32_t off = lane * pitch;
  for (int i = 0; i < iters; i++) {
    st<W>(base, off & (region - 1), (uint32_t)i);
    st<W>(base, (off + delta) & (region - 1), (uint32_t)i + 7);
    off += 64 * pitch;
  }
}

__global__ void k_atomic(uint32_t* buf, uint32_t region_words, uint32_t stride, int iters) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  uint32_t* base = buf + (size_t)wave * (region_words + 1024);
  uint32_t bit = lane * stride;
  for (int i = 0; i < iters; i++) {
    __hip_atomic_fetch_or(base + ((bit >> 5) & (region_words - 1)), 1u << (bit & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bit += 64 * stride;
  }
}

// dependent integer chain, CHAINS independent chains per lane
template <int CHAINS>
__global__ void k_valu(uint32_t* out, int iters) {
  uint32_t a[CHAINS];
  for (int c = 0; c < CHAINS; c++) a[c] = threadIdx.x + c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) a[c] = (a[c] ^ 0x9E3779B9u) + (a[c] >> 3);   // 3 dependent VALU (xor, shift, add) -> the compiler may fuse: count from ISA
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; c++) s += a[c];
  if (s == 0x12345u) out[0] = s;
}

// SALU next to VALU: a scalar dependent chain interleaved with the vector one
__global__ void k_salu_valu(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x, s = seed;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      a = (a ^ 0x9E3779B9u) + (a >> 3);
      s = (s ^ 0x7F4A7C15u) + (s >> 5);
      s = __builtin_amdgcn_readfirstlane(s);
    }
  }
  if (a + s == 0x12345u) out[0] = a;
}

int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  printf("CUs %d clock %.2f GHz\n", cus, ghz);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const uint32_t region = 4096;
  const int iters = 2000;
  uint8_t* d; (void)hipMalloc(&d, (size_t)cus * 32 * (region + 4096) + (1 << 20));
  (void)hipMemset(d, 0, (size_t)cus * 32 * (region + 4096));
  uint32_t* o; (void)hipMalloc(&o, 64);
  struct Case { int w; uint32_t pitch, mis, mod; const char* what; };
  const Case cases[] = {
      {4, 4, 0, 1, "dword coalesced"}, {8, 8, 0, 1, "qword coalesced"}, {16, 16, 0, 1, "x4 coalesced aligned"},
      {16, 16, 4, 1, "x4 pitch16 mis4"}, {16, 16, 1, 1, "x4 pitch16 mis1"}, {16, 20, 0, 1, "x4 pitch20 (strings)"},
      {16, 20, 3, 1, "x4 pitch20 mis3"}, {16, 33, 0, 1, "x4 pitch33"}, {8, 11, 0, 1, "qword pitch11"}, {8, 9, 3, 1, "qword pitch9 mis3"},
      {4, 5, 0, 1, "dword pitch5"}, {4, 7, 1, 1, "dword pitch7 mis1"}, {2, 3, 0, 1, "short pitch3"}, {1, 1, 0, 1, "byte pitch1"},
      {1, 3, 0, 1, "byte pitch3"}, {16, 20, 0, 2, "x4 pitch20 half lanes"}, {16, 20, 0, 4, "x4 pitch20 quarter lanes"},
      {16, 20, 0, 16, "x4 pitch20 4 lanes"}, {8, 8, 0, 64, "qword one lane (bitmap)"}, {4, 4, 0, 2, "dword coalesced half lanes"},
      {1, 1, 0, 1, "byte pitch1 (type ids)"}, {4, 120, 0, 1, "dword pitch120 (scatter)"}, {16, 16, 0, 1, "x4 coalesced aligned (again)"},
  };
  for (int wpc : {16, 8, 4}) {          // waves per CU
    const int blocks = cus * wpc / 4;
    printf("---- %d waves per CU\n", wpc);
    for (const Case& c : cases) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        switch (c.w) {
          case 1: hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, d, region, c.pitch, c.mis, c.mod, iters); break;
          case 2: hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, d, region, c.pitch, c.mis, c.mod, iters); break;
          case 4: hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(256), 0, 0, d, region, c.pitch, c.mis, c.mod, iters); break;
          case 8: hipLaunchKernelGGL(k_store<8>, dim3(blocks), dim3(256), 0, 0, d, region, c.pitch, c.mis, c.mod, iters); break;
          default: hipLaunchKernelGGL(k_store<16>, dim3(blocks), dim3(256), 0, 0, d, region, c.pitch, c.mis, c.mod, iters); break;
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double instr_per_cu = (double)wpc * iters;
      const double lanes = 64.0 / c.mod;
      printf("store w=%2d pitch=%3u mis=%u lanes=%2.0f %-28s: %.3f ms  %6.1f cycles/instr/CU  %7.1f GB/s useful\n", c.w, c.pitch, c.mis, lanes,
             c.what, best, best * 1e-3 * ghz * 1e9 / instr_per_cu, (double)blocks * 4 * iters * lanes * c.w / best / 1e6);
    }
    for (uint32_t delta : {4u, 8u, 64u, 2048u}) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_store_pair<16>, dim3(blocks), dim3(256), 0, 0, d, region, 20u, delta, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("store PAIR w=16 pitch=20, second store +%4u bytes: %.3f ms  %6.1f cycles/instr/CU\n", delta, best, best * 1e-3 * ghz * 1e9 / ((double)wpc * iters * 2));
      best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_store_pair<8>, dim3(blocks), dim3(256), 0, 0, d, region, 11u, delta, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("store PAIR w= 8 pitch=11, second store +%4u bytes: %.3f ms  %6.1f cycles/instr/CU\n", delta, best, best * 1e-3 * ghz * 1e9 / ((double)wpc * iters * 2));
    }
    for (uint32_t stride : {1u, 3u, 40u}) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, (uint32_t*)d, 1024u, stride, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("atomicOr bit stride %2u: %.3f ms  %6.1f cycles/instr/CU\n", stride, ms, ms * 1e-3 * ghz * 1e9 / ((double)wpc * iters));
    }
  }
  // VALU issue: waves per SIMD x chains
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = cus, threads = 64 * 4 * wps;   // one block per CU, wps waves on each SIMD
    const int it = 4000;
    float ms1, ms2, ms3;
    for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(threads), 0, 0, o, it); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms1, e0, e1); }
    for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(threads), 0, 0, o, it); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms2, e0, e1); }
    for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_salu_valu, dim3(blocks), dim3(threads), 0, 0, o, it, 77u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms3, e0, e1); }
    // per wave: it*16 rounds of (xor, shift, add) per chain
    const double rounds = (double)it * 16;
    printf("valu %d waves/SIMD: 1 chain %.3f ms = %.2f cycles/round/wave; 2 chains %.3f ms = %.2f cycles/round/wave; valu+salu chain %.3f ms = %.2f cycles/round/wave\n", wps, ms1,
           ms1 * 1e-3 * ghz * 1e9 / rounds, ms2, ms2 * 1e-3 * ghz * 1e9 / rounds, ms3, ms3 * 1e-3 * ghz * 1e9 / rounds);
  }
  return 0;
}
