// How fast can ONE wave per SIMD issue?  One workgroup on one CU, 1 / 2 / 4 waves per SIMD, each wave runs a loop of N independent v_sad_u8
// chains (ILP = 1, 2, 4, 8); reports cycles per wave-instruction.  hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP> __global__ void k(unsigned *out, unsigned long long *cyc, int iters)
{
  unsigned a[ILP];
  for (int k = 0; k < ILP; k++) a[k] = threadIdx.x + k;
  unsigned b = threadIdx.x * 0x01010101u, c = 0x03050709u;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int k = 0; k < ILP; k++) a[k] = __builtin_amdgcn_sad_u8(b, c + r, a[k]);
  }
  unsigned long long t1 = clock64();
  unsigned s = 0;
  for (int k = 0; k < ILP; k++) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP> void run(int threads)
{
  unsigned *out; unsigned long long *cyc, h;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<ILP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<ILP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d  ILP %d: %.2f cycles per wave-instruction (s_memtime ticks)\n", threads / 256, ILP, (double)h / (iters * 8.0 * ILP));
  hipFree(out); hipFree(cyc);
}
int main() { for (int t : {256, 512, 1024}) { run<1>(t); run<2>(t); run<4>(t); run<8>(t); } return 0; }
