// What a result's host memory costs on this box: hipHostMalloc, D2H into pinned / pageable (fresh, untouched) memory,
// hipHostRegister of a malloc'd range -- per block size.  hipcc -O2 tools/d2hcost.hip -o tools/d2hcost && tools/d2hcost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  for (size_t mb : {21, 170, 1700}) {
    const size_t n = mb << 20;
    void* d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n)); CK(hipDeviceSynchronize());
    void* warm; CK(hipHostMalloc(&warm, n, 0));
    CK(hipMemcpyAsync(warm, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    double t = now(); CK(hipMemcpyAsync(warm, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double pinned = now() - t;
    t = now(); void* h2; CK(hipHostMalloc(&h2, n, 0)); const double hm = now() - t;
    t = now(); CK(hipMemcpyAsync(h2, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double first = now() - t;
    t = now(); CK(hipHostFree(h2)); const double hf = now() - t;
    void* pg = nullptr; posix_memalign(&pg, 4096, n);
    t = now(); CK(hipMemcpyAsync(pg, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double page_cold = now() - t;
    t = now(); CK(hipMemcpyAsync(pg, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double page_warm = now() - t;
    free(pg);
    posix_memalign(&pg, 4096, n);
    t = now(); CK(hipHostRegister(pg, n, 0)); const double reg = now() - t;
    t = now(); CK(hipMemcpyAsync(pg, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double reg_copy = now() - t;
    t = now(); CK(hipHostUnregister(pg)); const double unreg = now() - t;
    free(pg);
    // staged: D2H into the warm pinned block, then a multi-threaded memcpy into fresh pageable memory
    posix_memalign(&pg, 4096, n);
    for (int nt : {8, 16, 32}) {
      t = now();
      std::vector<std::thread> th;
      for (int i = 0; i < nt; i++) th.emplace_back([&, i] { size_t a = n * i / nt, b = n * (i + 1) / nt; memcpy((char*)pg + a, (char*)warm + a, b - a); });
      for (auto& x : th) x.join();
      printf("  memcpy pinned->pageable %zu MB with %d threads: %.2f ms (%.1f GB/s)%s\n", mb, nt, now() - t, n / (now() - t) / 1e6, nt == 8 ? " [cold pages]" : "");
    }
    free(pg);
    printf("%4zu MB: D2H->pinned(warm) %.2f ms (%.1f GB/s) | hipHostMalloc %.2f ms, first D2H into it %.2f ms, hipHostFree %.2f | D2H->pageable cold %.2f ms (%.1f GB/s) warm %.2f | "
           "hipHostRegister %.2f ms + D2H %.2f + unregister %.2f\n", mb, pinned, n / pinned / 1e6, hm, first, hf, page_cold, n / page_cold / 1e6, page_warm, reg, reg_copy, unreg);
    {   // both directions at once, pinned on both sides, two streams: is the link full duplex for the copy engines?
      hipStream_t st2; CK(hipStreamCreate(&st2));
      void* up; CK(hipHostMalloc(&up, n, 0)); memset(up, 2, n);
      void* d2; CK(hipMalloc(&d2, n));
      CK(hipMemcpyAsync(d2, up, n, hipMemcpyHostToDevice, st2)); CK(hipStreamSynchronize(st2));
      t = now(); CK(hipMemcpyAsync(d2, up, n, hipMemcpyHostToDevice, st2)); CK(hipStreamSynchronize(st2)); const double h2d_alone = now() - t;
      t = now();
      CK(hipMemcpyAsync(d2, up, n, hipMemcpyHostToDevice, st2));
      CK(hipMemcpyAsync(warm, d, n, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st2)); const double h2d_both = now() - t;
      CK(hipStreamSynchronize(st)); const double both = now() - t;
      printf("  duplex %zu MB each way: H2D alone %.2f ms (%.1f GB/s), D2H alone %.2f ms; together: H2D done at %.2f ms, both at %.2f ms = %.1f GB/s combined\n",
             mb, h2d_alone, n / h2d_alone / 1e6, pinned, h2d_both, both, 2.0 * n / both / 1e6);
      CK(hipFree(d2)); CK(hipHostFree(up)); CK(hipStreamDestroy(st2));
    }
    CK(hipHostFree(warm)); CK(hipFree(d));
  }
  return 0;
}
