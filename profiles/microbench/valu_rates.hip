// valu_rates.hip -- measures issue rates of the integer VALU instructions the ME kernels are built from,
// on the GPU it runs on (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
// Output: one line per instruction: lane-ops per clock per CU (64-wide wave on a SIMD-32 = 2 clk => 128/CU peak
// for a full-rate op with 4 SIMDs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define CHAINS 8

#define KERNEL(name, DECL, BODY)                                                        \
  __global__ __launch_bounds__(256) void k_##name(uint32_t *out, uint32_t seed) {        \
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 12345u;  \
    DECL                                                                                 \
    for (int i = 0; i < ITERS; i++) { BODY }                                             \
    uint32_t r = 0;                                                                      \
    for (int k = 0; k < CHAINS; k++) r ^= (uint32_t)x[k];                                 \
    if (r == 0x12345678u) out[threadIdx.x] = r ^ a ^ b ^ c;                              \
  }

// 32-bit accumulator chains
#define D32 uint32_t x[CHAINS]; for (int k = 0; k < CHAINS; k++) x[k] = a + k;
#define D64 uint64_t x[CHAINS]; for (int k = 0; k < CHAINS; k++) x[k] = ((uint64_t)a << 32) | (b + k);
#define REP(stmt) _Pragma("unroll") for (int k = 0; k < CHAINS; k++) { stmt }

KERNEL(v_sad_u8,        D32, REP(asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_sad_u8_sgpr,   D32, REP(asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x[k]) : "s"(seed), "v"(b));))
KERNEL(v_sad_hi_u8_sgpr, D32, REP(asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(x[k]) : "s"(seed), "v"(b));))
KERNEL(v_sad_u16,       D32, REP(asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_msad_u8,       D32, REP(asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_alignbyte,     D32, REP(asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(c));))
KERNEL(v_add_u32,       D32, REP(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));))
KERNEL(v_add3_u32,      D32, REP(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_lshl_add_u32,  D32, REP(asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(x[k]) : "v"(a));))
KERNEL(v_min_u32,       D32, REP(asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));))
KERNEL(v_min3_u32,      D32, REP(asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_mad_u32_u16,   D32, REP(asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_mad_u32_u24,   D32, REP(asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));))
KERNEL(v_pk_add_u16,    D32, REP(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[k]) : "v"(a));))
KERNEL(v_pk_min_u16,    D32, REP(asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[k]) : "v"(a));))
KERNEL(v_perm_b32,      D32, REP(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(c));))
KERNEL(v_qsad_pk_u16_u8,  D64, REP(asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(x[k]) : "v"(((uint64_t)a << 32) | b), "v"(c));))
KERNEL(v_mqsad_pk_u16_u8, D64, REP(asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(x[k]) : "v"(((uint64_t)a << 32) | b), "v"(c));))
KERNEL(v_min_f64,       D64, REP(asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[k]) : "v"(((uint64_t)(a | 0x40000000u) << 32) | b));))
KERNEL(v_cmp_lt_u64_cnd, D64, REP(x[k] = (((uint64_t)a << 32) | (b + i)) < x[k] ? (((uint64_t)a << 32) | (b + i)) : x[k];))

// shader clock (s_memtime) against the 100 MHz wall clock (s_memrealtime) while every CU is busy with v_sad_u8
__global__ __launch_bounds__(256) void k_clock(unsigned long long *out, uint32_t seed)
{
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint32_t x[CHAINS]; for (int k = 0; k < CHAINS; k++) x[k] = a + k;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < ITERS; i++) { REP(asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));) }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  uint32_t r = 0; for (int k = 0; k < CHAINS; k++) r ^= x[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = r; }
}

template <typename F>
static void run(const char *name, F kern, double ops_per_inst)
{
  uint32_t *d; hipMalloc(&d, 1024);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount, blocks = cus * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u + rep); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double insts = (double)blocks * 256 * ITERS * CHAINS;          // lane-instructions
  double clk = p.clockRate * 1e3;                                 // Hz (max)
  double lane_ops_per_clk_cu = insts / (best * 1e-3) / clk / cus;
  printf("%-20s %8.3f ms  %7.1f lane-inst/clk/CU (at %d MHz nominal)  %8.2f T lane-inst/s  x%.0f = %8.2f T elem-ops/s\n", name, best,
         lane_ops_per_clk_cu, p.clockRate / 1000, insts / (best * 1e-3) / 1e12, ops_per_inst, insts / (best * 1e-3) / 1e12 * ops_per_inst);
  hipFree(d);
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s  CUs %d  clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
{
    unsigned long long *d, h[3]; hipMalloc(&d, 64);
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_clock, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("clock calibration: %llu s_memtime ticks in %llu wall ticks (100 MHz) -> s_memtime runs at %.0f MHz; 8 waves/SIMD x %d x %d v_sad_u8 -> %.2f ticks per wave-instruction\n",
             h[0], h[1], (double)h[0] / h[1] * 100.0, ITERS, CHAINS, (double)h[0] / (8.0 * ITERS * CHAINS) * 4 / 4);
    }
    hipFree(d);
  }
#define RUN(n, ops) run(#n, k_##n, ops)
  RUN(v_add_u32, 1); RUN(v_add3_u32, 1); RUN(v_lshl_add_u32, 1); RUN(v_min_u32, 1); RUN(v_min3_u32, 2);
  RUN(v_sad_u8, 4); RUN(v_sad_u8_sgpr, 4); RUN(v_sad_hi_u8_sgpr, 4); RUN(v_sad_u16, 2); RUN(v_msad_u8, 4); RUN(v_alignbyte, 1); RUN(v_perm_b32, 1);
  RUN(v_mad_u32_u16, 1); RUN(v_mad_u32_u24, 1); RUN(v_pk_add_u16, 2); RUN(v_pk_min_u16, 2);
  RUN(v_qsad_pk_u16_u8, 16); RUN(v_mqsad_pk_u16_u8, 16); RUN(v_min_f64, 1); RUN(v_cmp_lt_u64_cnd, 1);
  return 0;
}
