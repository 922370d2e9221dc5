// residency probe: N workgroups of 384 threads, 168 VGPRs (launch bounds), L bytes of dynamic LDS: how many run at the same time?
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ char sm[];
__global__ __launch_bounds__(384, 3) void k(unsigned *cnt, unsigned *out, int target)
{
  if (threadIdx.x == 0) {
    atomicAdd(cnt, 1u);
    unsigned long long t0 = wall_clock64();
    unsigned seen = 0;
    while (wall_clock64() - t0 < 20000000ull) { seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((int)seen >= target) break; __builtin_amdgcn_s_sleep(10); }
    out[blockIdx.x] = seen;
    sm[0] = 1;
  }
}
int main() {
  unsigned *cnt, *out; hipMalloc(&cnt, 4); hipMalloc(&out, 4096 * 4);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  for (int lds : {61440, 65536, 80000}) for (int n : {256, 512, 768}) {
    hipMemset(cnt, 0, 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(384), lds, 0, cnt, out, n);
    hipDeviceSynchronize();
    unsigned h[1024]; hipMemcpy(h, out, n * 4, hipMemcpyDeviceToHost);
    unsigned mn = ~0u, mx = 0; for (int i = 0; i < n; i++) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 384, lds);
    printf("lds %d grid %d: workgroups seen at once min %u max %u (runtime's count per CU: %d)\n", lds, n, mn, mx, occ);
  }
  return 0;
}
