// microbench: throughput of per-lane 8-byte global stores/loads, aligned vs unaligned, packed-contiguous per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t __attribute__((aligned(1))) u64u;
template <int STORE>
__global__ void k(uint8_t* buf, uint32_t mis, uint32_t pitch, int iters, uint64_t* sink) {
  // lane i of block b touches 8 bytes at base + i*pitch + mis; each iteration moves on by 64*pitch
  uint8_t* p = buf + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * pitch + mis;
  const size_t step = (size_t)gridDim.x * blockDim.x * pitch;
  uint64_t acc = 0;
  for (int i = 0; i < iters; i++) {
    if (STORE) *(u64u*)p = (uint64_t)i + threadIdx.x;
    else acc += *(const u64u*)p;
    p += step;
  }
  if (!STORE && acc == 0x123456789ull) sink[0] = acc;
}
int main() {
  const size_t bytes = 2ull << 30;
  uint8_t* d; uint64_t* s; (void)hipMalloc(&d, bytes + 4096); (void)hipMalloc(&s, 8);
  (void)hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int store = 1; store >= 0; store--)
    for (uint32_t pitch : {8u, 9u, 11u, 24u})
      for (uint32_t mis : {0u, 3u}) {
        const int blocks = 2048, threads = 256;
        const int iters = (int)(bytes / ((size_t)blocks * threads * pitch));
        for (int rep = 0; rep < 2; rep++) {
          (void)hipEventRecord(e0);
          if (store) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, mis, pitch, iters, s);
          else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, mis, pitch, iters, s);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double useful = (double)blocks * threads * iters * 8;
        printf("%s pitch=%2u mis=%u : %.3f ms  %.0f GB/s useful (%.2f G lane-ops/s)\n", store ? "store" : "load ", pitch, mis, ms, useful / ms / 1e6, useful / 8 / ms / 1e6);
      }
  return 0;
}
