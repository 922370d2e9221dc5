// v_ashr_pk_u8_i32 on gfx950: what the instruction returns, against clamp(x >> s, 0, 255), and what it leaves in the upper half of its
// destination.  hipcc (ROCm 7.2) fuses pairs of `min(max(x >> s, 0), 255)` into this instruction and then ORs further bytes into bits 16..31
// of the result as if they were zero (seen in k_intra_chroma, jm_amd/csrc/intra.hip, where the destination register was also the first
// source: a negative source left 0xffff there).   build: hipcc --offload-arch=gfx950 -O2 ashr_pk_u8.hip -o ashr_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const int *x, int n, int s, unsigned *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned r = 0xabcd0000u;                               // what was in the destination register before
  const int a = x[i], b = x[i];
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "+v"(r) : "v"(a), "v"(b), "v"(s));
  out[i] = r;
}

int main()
{
  const int lo = -70000, hi = 70000, n = hi - lo + 1, s = 5;
  std::vector<int> h(n);
  for (int i = 0; i < n; i++) h[i] = lo + i;
  int *dx; unsigned *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, s, dout);
  std::vector<unsigned> o(n);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int bad = 0, first_bad = 0, last_bad = 0;
  for (int i = 0; i < n; i++) {
    const int v = h[i] >> s, want = v < 0 ? 0 : (v > 255 ? 255 : v);
    const int got0 = o[i] & 255, got1 = (o[i] >> 8) & 255;
    if (got0 != want || got1 != want) {
      if (!bad) first_bad = h[i];
      last_bad = h[i];
      if (bad < 6 || (h[i] > -900 && h[i] < -700 && (h[i] & 31) == 0)) printf("x = %d (x >> %d = %d): instruction gives %d / %d, clamp gives %d\n", h[i], s, v, got0, got1, want);
      bad++;
    }
  }
  printf("%d of %d inputs in [%d, %d] differ from clamp(x >> %d, 0, 255); first %d, last %d\n", bad, n, lo, hi, s, first_bad, last_bad);
  int kept = 0, zeroed = 0, other = 0;
  for (int i = 0; i < n; i++) { const unsigned up = o[i] >> 16; if (up == 0xabcd) kept++; else if (up == 0) zeroed++; else other++; }
  printf("bits 16..31 of the destination afterwards: kept the old value 0xabcd in %d cases, zero in %d, something else in %d\n", kept, zeroed, other);
  return 0;
}
