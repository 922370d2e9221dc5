// me_common.h -- helpers shared by the integer-pel ME kernels (me_fullsearch.hip, me_fast.hip).
#pragma once
#include "jmhip_internal.h"

#define NP JMHIP_NPART

#define JMHIP_HAVE_IABS 1
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
// mvbits LUT of lencod/src/mv_search.c:366-374 in closed form
__device__ __forceinline__ int mvbits(int d)
{
  int a = iabs_(d);
  return a == 0 ? 1 : 2 * (31 - __clz(a)) + 3;
}
// index of (dx,dy) in JM's spiral (mv_search.c:405-442)
__device__ __forceinline__ int spiral_index(int dx, int dy)
{
  int ax = iabs_(dx), ay = iabs_(dy), l = imax_(ax, ay);
  if (l == 0) return 0;
  int base = (2 * l - 1) * (2 * l - 1);
  if (ay == l && ax < l) return base + 2 * (dx + l - 1) + (dy > 0);
  return base + 2 * (2 * l - 1) + 2 * (dy + l) + (dx > 0);
}

// sixteen 4x4 SADs of the macroblock at window position (wx, wy); s_win rows are wpitch bytes;
// cur: the 16x16 current macroblock as 64 dwords (LDS pointer or register array)
__device__ __forceinline__ void sad16(const uint8_t *s_win, int wpitch, const uint32_t *cur, int wx, int wy, uint32_t s7[16])
{
#pragma unroll
  for (int k = 0; k < 16; k++) s7[k] = 0;
  const int sh = wx & 3;
  const uint32_t *row = (const uint32_t *)(s_win + wy * wpitch + (wx & ~3));
  const int wp4 = wpitch >> 2;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    uint32_t a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3], a4 = row[4];
    uint32_t b0 = __builtin_amdgcn_alignbyte(a1, a0, sh);
    uint32_t b1 = __builtin_amdgcn_alignbyte(a2, a1, sh);
    uint32_t b2 = __builtin_amdgcn_alignbyte(a3, a2, sh);
    uint32_t b3 = __builtin_amdgcn_alignbyte(a4, a3, sh);
    const int q = (r >> 2) * 4;
    s7[q + 0] = __builtin_amdgcn_sad_u8(b0, cur[r * 4 + 0], s7[q + 0]);
    s7[q + 1] = __builtin_amdgcn_sad_u8(b1, cur[r * 4 + 1], s7[q + 1]);
    s7[q + 2] = __builtin_amdgcn_sad_u8(b2, cur[r * 4 + 2], s7[q + 2]);
    s7[q + 3] = __builtin_amdgcn_sad_u8(b3, cur[r * 4 + 3], s7[q + 3]);
    row += wp4;
  }
}

// the 41 partition SADs in the ABI's partition order (jmhip.h); update_full_search_large_blocks, me_fullfast.c:195-260
__device__ __forceinline__ void aggregate41(const uint32_t s7[16], uint32_t sp[NP])
{
#pragma unroll
  for (int k = 0; k < 16; k++) sp[25 + k] = s7[k];                                        // 4x4
#pragma unroll
  for (int by = 0; by < 4; by++) { sp[9 + by * 2] = s7[by * 4] + s7[by * 4 + 1]; sp[9 + by * 2 + 1] = s7[by * 4 + 2] + s7[by * 4 + 3]; }   // 8x4
#pragma unroll
  for (int bx = 0; bx < 4; bx++) { sp[17 + bx] = s7[bx] + s7[4 + bx]; sp[21 + bx] = s7[8 + bx] + s7[12 + bx]; }                             // 4x8
  sp[5] = sp[9] + sp[11];  sp[6] = sp[10] + sp[12];  sp[7] = sp[13] + sp[15];  sp[8] = sp[14] + sp[16];                                     // 8x8
  sp[1] = sp[5] + sp[6];   sp[2] = sp[7] + sp[8];                                                                                           // 16x8
  sp[3] = sp[5] + sp[7];   sp[4] = sp[6] + sp[8];                                                                                           // 8x16
  sp[0] = sp[1] + sp[2];                                                                                                                    // 16x16
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(k, off, 64);
    k = o < k ? o : k;
  }
  return k;
}

// Which kernel owns a job: the tuned kernel (me_fast.hip) takes search ranges up to 32 whose 32-bit
// (cost << 11 | tag) key cannot overflow and whose max_mvd guard (me_fullfast.c:638,671) cannot trigger
// anywhere in the window; everything else goes to the generic kernel.  Wave-uniform.
// Key bound: an mv component difference is < 2^17, so mvbits <= 35 per component; lambda < 14000 keeps the rate below 2^20,
// and with SAD << 5 <= 32640 * 32 (every partition but 16x16, which has one more bit) the cost stays below 2^21.
__device__ __forceinline__ bool job_is_fast(const jmhip_me_job *__restrict__ job)
{
  const int R = job->search_range;
  if (R > 32 || R < 1 || job->lambda < 0 || job->lambda >= 14000) return false;
  if (job->max_mvd != 0) {
    const int guard = job->max_mvd - 1;
    const uint64_t mask = job->part_mask;
    for (int p = 0; p < NP; p++)
      if ((mask >> p) & 1) {
        const int m = imax_(iabs_(job->center_x - job->pred[p][0]), iabs_(job->center_y - job->pred[p][1]));
        if (m + 4 * R >= guard) return false;
      }
  }
  return true;
}

// Workgroup b runs on XCD b % 8 (observed placement, used for speed only): give every XCD a CONTIGUOUS run of jobs, so that jobs
// whose search windows / candidate blocks share cache lines (horizontal neighbours) also share an L2.  A bijection of [0, n).
__device__ __forceinline__ int xcd_job_index(int b, int n)
{
  const int k = b & 7, q = n >> 3, r = n & 7;
  return k * q + (k < r ? k : r) + (b >> 3);
}

void jmhip_launch_me_fast(jmhip_ctx *ctx, int slot, const jmhip_me_job *d_jobs, int njobs, jmhip_me_result *d_results, unsigned *d_declined);
