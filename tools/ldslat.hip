// microbench: dependent-chain LDS read latency, aligned vs unaligned 8-byte reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;

template <int MODE>
__global__ void chase(uint64_t* out, int steps, uint32_t stride, uint32_t mis) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // fill: every 8-byte slot holds the byte offset of the next slot to visit (per lane ring)
  const uint32_t nslots = 4096;
  for (uint32_t i = threadIdx.x; i < nslots * 2; i += blockDim.x) ((uint32_t*)lds)[i] = 0;
  __syncthreads();
  // each lane walks: pos -> pos + stride (mod), stored value = next pos
  for (uint32_t i = threadIdx.x; i < nslots; i += blockDim.x) {
    uint32_t pos = (i * 8 + mis);
    uint32_t nxt = (((i + stride) % (nslots - 2)) * 8 + mis);
    *(u32u*)(lds + pos) = nxt;
  }
  __syncthreads();
  uint32_t p = threadIdx.x * 8 + mis;
  uint64_t acc = 0;
  long long t0 = clock64();
  for (int s = 0; s < steps; s++) {
    if (MODE == 0) { p = *(uint32_t*)(lds + (p & ~3u)); }            // aligned b32 (only valid when mis==0)
    else if (MODE == 1) { uint64_t v = *(u64u*)(lds + p); p = (uint32_t)v; acc += v >> 32; }   // b64, alignment = mis
    else { uint32_t v = *(u32u*)(lds + p); p = v; }                  // b32 unaligned-capable
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (uint64_t)(t1 - t0);
  if (p == 0xFFFFFFFF) out[0] = acc;
}

int main() {
  uint64_t* d; hipMalloc(&d, 8 * 4096);
  const int steps = 2000;
  auto run = [&](int mode, int blocks, int threads, uint32_t stride, uint32_t mis, const char* name) {
    hipMemset(d, 0, 8 * 4096);
    if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(threads), 40000, 0, d, steps, stride, mis);
    if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(threads), 40000, 0, d, steps, stride, mis);
    if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(threads), 40000, 0, d, steps, stride, mis);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= blocks;
    printf("%-28s blocks=%5d thr=%4d stride=%4u mis=%u : %.1f cycles/step\n", name, blocks, threads, stride, mis, s / steps);
  };
  for (int blocks : {1, 1024}) {
    for (int threads : {64, 256}) {
      run(0, blocks, threads, 1, 0, "b32 aligned seq");
      run(1, blocks, threads, 1, 0, "b64 aligned seq");
      run(1, blocks, threads, 1, 1, "b64 mis=1 seq");
      run(1, blocks, threads, 1, 4, "b64 mis=4 seq");
      run(1, blocks, threads, 15, 0, "b64 aligned stride15");
      run(1, blocks, threads, 15, 3, "b64 mis=3 stride15");
      run(2, blocks, threads, 15, 3, "b32 mis=3 stride15");
      run(2, blocks, threads, 15, 0, "b32 aligned stride15");
    }
  }
  return 0;
}
