// How long does the host take to learn that the last kernel of a call has finished?  (MI355X; DESIGN.md "per-call cost")
//   A  kernel + hipStreamSynchronize
//   B  kernel + hipMemcpyAsync(512 B, D2H, pinned) + hipStreamSynchronize          (what a call did until round 3)
//   C  kernel that stores 512 B + a flag into host-mapped pinned memory (system scope) + host spin on the flag
//   D  as C, but the flag is written by a second tiny kernel behind a 100 us kernel (the in-order stream does the ordering)
//   E  kernel + hipEventRecord + spin on hipEventQuery
// Build: hipcc --offload-arch=gfx950 -O2 tools/synclat.hip -o tools/synclat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <atomic>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long cycles, uint32_t* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0) sink[0] = 1;
}
__global__ void publish(const uint32_t* src, uint32_t* host, uint32_t words, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) __builtin_nontemporal_store(src[i], host + 16 + i);
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *dsrc, *hbuf, *hdev; void* hp;
  CK(hipMalloc(&dsrc, 4096)); CK(hipMemset(dsrc, 1, 4096));
  CK(hipHostMalloc(&hp, 4096, hipHostMallocMapped)); hbuf = (uint32_t*)hp; std::memset(hbuf, 0, 4096);
  CK(hipHostGetDevicePointer((void**)&hdev, hp, 0));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const unsigned long long k100us = 100ull * 100;   // wall_clock64 ticks at 100 MHz
  const int R = 200;
  for (int mode = 0; mode < 5; mode++) {
    double tot = 0, kern = 0;
    for (int r = -20; r < R; r++) {
      const double t0 = now_us();
      switch (mode) {
        case 0: hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, k100us, nullptr); CK(hipStreamSynchronize(s)); break;
        case 1: hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, k100us, nullptr);
                CK(hipMemcpyAsync(hbuf + 64, dsrc, 512, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); break;
        case 2: { const uint32_t seq = (uint32_t)(r + 100);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, k100us, nullptr);
                hipLaunchKernelGGL(publish, dim3(1), dim3(128), 0, s, dsrc, hdev, 128u, seq);
                while (__atomic_load_n((volatile uint32_t*)hbuf, __ATOMIC_ACQUIRE) != seq) {} break; }
        case 3: { const uint32_t seq = (uint32_t)(r + 1000);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, k100us, nullptr);
                hipLaunchKernelGGL(publish, dim3(1), dim3(128), 0, s, dsrc, hdev, 128u, seq);
                while (__atomic_load_n((volatile uint32_t*)hbuf, __ATOMIC_ACQUIRE) != seq) { __builtin_ia32_pause(); } CK(hipStreamQuery(s) == hipErrorNotReady ? hipSuccess : hipSuccess); break; }
        case 4: hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, k100us, nullptr); CK(hipEventRecord(ev, s));
                while (hipEventQuery(ev) == hipErrorNotReady) {} break;
      }
      const double t1 = now_us();
      if (r >= 0) tot += t1 - t0;
    }
    static const char* names[] = {"A kernel + hipStreamSynchronize", "B kernel + 512 B D2H copy + hipStreamSynchronize",
                                  "C kernel + publish kernel -> mapped host memory, host spins on the flag", "D as C with pause in the spin",
                                  "E kernel + event record + spin on hipEventQuery"};
    std::printf("%-80s %8.2f us per call (kernel itself spins 100 us)\n", names[mode], tot / R);
  }
  // back-to-back calls: how much GPU idle time between the end of one call's last op and the start of the next call's kernel
  return 0;
}
