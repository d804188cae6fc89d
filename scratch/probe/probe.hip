#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_add(const int* a, int* b, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) b[i]=a[i]+1; }
extern "C" int probe_info(){
  int n=0; hipError_t e=hipGetDeviceCount(&n); if(e!=hipSuccess){printf("devcount err %s\n",hipGetErrorString(e));return -1;}
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  int rv=0; hipRuntimeGetVersion(&rv);
  printf("ndev=%d name=%s arch=%s cus=%d lds/block=%zu maxlds_optin=%zu rt=%d\n",n,p.name,p.gcnArchName,p.multiProcessorCount,p.sharedMemPerBlock,p.sharedMemPerBlockOptin,rv);
  return n;
}
// run kernel on caller-provided device pointers + stream
extern "C" int probe_run(const int* a, int* b, int n, void* stream){
  hipLaunchKernelGGL(k_add, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, a, b, n);
  hipError_t e=hipGetLastError(); if(e!=hipSuccess){printf("launch err %s\n",hipGetErrorString(e));return -1;}
  return 0;
}
extern "C" int probe_self(){
  int n=1000; int *a,*b; hipMalloc(&a,n*4); hipMalloc(&b,n*4); hipMemset(a,0,n*4);
  probe_run(a,b,n,nullptr); int h[1000]; hipMemcpy(h,b,n*4,hipMemcpyDeviceToHost);
  hipFree(a);hipFree(b); return h[999];
}
