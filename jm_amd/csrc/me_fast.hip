// me_fast.hip -- K1+K2+K3, tuned path for search ranges <= 32 (gfx950).
//
// Same result, bit for bit, as k_me_fullsearch (me_fullsearch.hip) -- i.e. as JM's
// full_search_motion_estimation (lencod/src/me_fullsearch.c:39-103) /
// fast_full_search_motion_estimation (lencod/src/me_fullfast.c:618-689) over the window, all 41 partitions --
// but organised around what the MI355X VALU can issue (profiles/r01_valu_rates.txt):
//   v_sad_hi_u8     4 abs-diff per lane-op at the full VOP3 rate (the SAD roofline: ~148 T abs-diff/s), result added at bit 16
//   v_add_u32       the fastest VALU op measured (87 lane-ops/clk/CU against 54-60 for the three-operand forms)
//   v_min3_u32      1 op folds TWO candidate keys into the running minimum
//   64-bit integer compare+select is 4.5x slower, v_alignbyte costs as much as the SAD it feeds.
//
// Mapping.  lane <-> window column (dx); a wave walks window rows two at a time; four waves share one job and four jobs fit a
// CU's LDS (37 KB each), so a CU holds 16 waves in four independent phases: one job's staging / merging overlaps the others'
// VALU-bound main pass.
//   * the search window is staged in LDS as FOUR byte-shifted copies, so the lane at column wx reads its 16 reference
//     bytes of a row as four aligned dwords from copy (wx & 3): no v_alignbyte in the main pass, and the copy stride
//     (== 16 dwords mod 64) spreads the 64 lanes of a read over all 64 banks;
//   * the current macroblock is fetched with scalar loads and stays in SGPRs: it is the scalar operand of v_sad_hi_u8;
//   * key layout.  The sixteen 4x4 accumulators START at rank (the position's order inside its column, below) and
//     v_sad_hi_u8 adds SAD << 16 on top; the 25 aggregation adds then give, for a partition of n 4x4 blocks,
//     (SAD << 16) + n * rank.  One v_add_u32 with the row constant (lambda * mvbits_y) << 11 makes it
//     ((SAD << 5) + rate) << 11 + n * rank = (cost << 11) | tag: cost in JM's distblk units, tag < 2^11 ordering the rows
//     of one column exactly as JM's spiral does (mv_search.c:405-442), so the unsigned minimum reproduces "first spiral
//     index wins".  The 16x16 partition (SAD up to 65280) is shifted right by one first: (cost << 10) | 8 * rank.
//     The y rate is uniform per row (a per-job LDS table, read as a broadcast b128), the x rate is constant per lane and
//     is added once at the end;
//   * two rows per step share a column, so one v_min3_u32 retires two candidates.
// ~1.2 VALU ops per (position, partition) instead of ~12 in the generic kernel.
//
// 2R+1 rows are odd: the last step takes rows (2R-1, 2R), re-evaluating row 2R-1 (a duplicate key cannot change a minimum).
// For R = 32 the 65th column is one more step of one wave: lane j evaluates rows (2j, 2j+1) of that column.
// Per-lane minima are merged across the waves (same lane == same column, keys comparable); the final stage turns the
// few candidates that attain the minimum cost into (cost, spiral index) pairs.
// All 41 partitions are always evaluated (no branches in the hot loop); part_mask selects what is stored.
//
// Jobs this kernel cannot take (search_range > 32, a max_mvd guard that could trigger, lambda too large
// for the 32-bit key) are left to k_me_fullsearch: both kernels evaluate job_is_fast() and exactly one
// of them processes a job.
#include "jmhip_internal.h"
#include "me_common.h"

// Phase profiler (profiles/prof_me.py builds a library with -DME_PROF): lane 0 of every wave stores the s_memtime ticks it spent
// in each phase; the time stamps sit after the scalar loads of the current macroblock (an earlier one would keep the compiler
// from using scalar loads at all).  Not compiled into the product library.
#ifdef ME_PROF
#define PROF_JOBS 8192
__device__ unsigned g_me_prof[PROF_JOBS * 8 * 8];
#define PROF_START unsigned long long tprev = clock64()
#define PROF(i) do { if (lane == 0) { const unsigned long long t_ = clock64(); g_me_prof[(blockIdx.x * 8 + wave) * 8 + (i)] = (unsigned)(t_ - tprev); tprev = t_; } } while (0)
extern "C" void jmhip_debug_read_me_prof(unsigned *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_prof), sizeof(g_me_prof)); }
#else
#define PROF_START
#define PROF(i)
#endif

#define WROWS_MAX 80                        // 2*32 + 16
#define CPITCH 20                           // dwords per row of one shifted copy (columns 0..64 + 15 -> 80 bytes)
#define NCOPY 4
#define CSTRIDE (WROWS_MAX * CPITCH + 16)   // dwords between copies: == 16 (mod 64) banks
#define COPY_DWORDS (NCOPY * CSTRIDE)
#define RYP 44                              // row pitch (dwords) of the y-rate table: 41 partitions padded to 11 x uint4
#define NW 4                                // waves per job
#define NT (NW * 64)
#define NWH (NW / 2)                        // the waves' minima are merged in two rounds so the merge area fits into the copy area
#define MPITCH 68                           // row pitch (dwords) of the published minima: == 4 (mod 64): 16 partitions x 4 threads hit 64 banks
#define MERGE_DWORDS (NWH * NP * MPITCH)
#define EXT_PITCH 34                        // 65th column: R + 1 row pairs per partition (overlays the y-rate table after the main pass)

static_assert(MERGE_DWORDS <= COPY_DWORDS, "merge area must fit into the window copies");
static_assert(NP * EXT_PITCH <= 65 * RYP, "65th-column keys must fit into the y-rate table");

__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
  unsigned r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ unsigned add_sat(unsigned a, unsigned b)
{
  unsigned r;
  asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// key layout per partition: cost sits above COST_SHIFT(p), the tag below it is rank << TAG_SHIFT(p) (n = 1 << TAG_SHIFT blocks)
__device__ __forceinline__ int cost_shift(int p) { return p == 0 ? 10 : 11; }
__device__ __forceinline__ int tag_shift(int p) { return p < 5 ? 3 : (p < 9 ? 2 : (p < 25 ? 1 : 0)); }

// 16 4x4 SADs (<< 16, on top of the initial values) for two vertically adjacent positions (rows wy, wy+1) of one column;
// base = dword index of the column's first window dword in its shifted copy
__device__ __forceinline__ void sad_pair(const uint32_t *s_mem, int base, int wy, const uint32_t (&cw)[64], unsigned (&sa)[16], unsigned (&sb)[16])
{
  int o = base + wy * CPITCH;
#pragma unroll
  for (int r = 0; r < 17; r++) {
    const uint32_t d0 = s_mem[o], d1 = s_mem[o + 1], d2 = s_mem[o + 2], d3 = s_mem[o + 3];
    if (r < 16) {
      const int q = (r >> 2) * 4;
      sa[q + 0] = __builtin_amdgcn_sad_hi_u8(d0, cw[r * 4 + 0], sa[q + 0]);
      sa[q + 1] = __builtin_amdgcn_sad_hi_u8(d1, cw[r * 4 + 1], sa[q + 1]);
      sa[q + 2] = __builtin_amdgcn_sad_hi_u8(d2, cw[r * 4 + 2], sa[q + 2]);
      sa[q + 3] = __builtin_amdgcn_sad_hi_u8(d3, cw[r * 4 + 3], sa[q + 3]);
    }
    if (r >= 1) {
      const int q = ((r - 1) >> 2) * 4;
      sb[q + 0] = __builtin_amdgcn_sad_hi_u8(d0, cw[(r - 1) * 4 + 0], sb[q + 0]);
      sb[q + 1] = __builtin_amdgcn_sad_hi_u8(d1, cw[(r - 1) * 4 + 1], sb[q + 1]);
      sb[q + 2] = __builtin_amdgcn_sad_hi_u8(d2, cw[(r - 1) * 4 + 2], sb[q + 2]);
      sb[q + 3] = __builtin_amdgcn_sad_hi_u8(d3, cw[(r - 1) * 4 + 3], sb[q + 3]);
    }
    o += CPITCH;
  }
}

// rank of row dy inside column |dx| = a, in JM's spiral order: 0 .. 2R.
// |dy| <= a: the column entries of ring a come in ascending dy; |dy| > a: ring |dy|, (dx,-|dy|) before (dx,+|dy|).
__device__ __forceinline__ unsigned col_rank(int dy, int a)
{
  const int u = dy < 0 ? -dy : dy;
  return (unsigned)(u <= a ? dy + a : 2 * u - 1 + (dy > 0));
}
__device__ __forceinline__ int rank_to_dy(unsigned rank, int a)
{
  if ((int)rank <= 2 * a) return (int)rank - a;
  const int t = (int)rank + 1, u = t >> 1;
  return (t & 1) ? u : -u;
}

// one step: rows (wy, wy+1) of the column whose window dwords start at `base`; keys folded into acc
__device__ __forceinline__ void step_pair(const uint32_t *s_mem, const uint32_t *s_ry, int base, int wy, int R, int a, const uint32_t (&cw)[64], unsigned (&acc)[NP])
{
  unsigned sa[16], sb[16], pa[NP], pb[NP];
  const unsigned rkA = col_rank(wy - R, a), rkB = col_rank(wy + 1 - R, a);
#pragma unroll
  for (int k = 0; k < 16; k++) { sa[k] = rkA; sb[k] = rkB; }
  sad_pair(s_mem, base, wy, cw, sa, sb);
  aggregate41(sa, pa);
  aggregate41(sb, pb);
  const uint4 *ryA = (const uint4 *)(s_ry + wy * RYP), *ryB = (const uint4 *)(s_ry + (wy + 1) * RYP);
#pragma unroll
  for (int g = 0; g < 11; g++) {
    const uint4 ca = ryA[g], cb = ryB[g];                 // broadcast reads: 4 partitions' row constants
    const unsigned cav[4] = {ca.x, ca.y, ca.z, ca.w}, cbv[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int p = 4 * g + t;
      if (p == 0) acc[p] = umin3(acc[p], (pa[p] >> 1) + cav[t], (pb[p] >> 1) + cbv[t]);
      else if (p < NP) acc[p] = umin3(acc[p], pa[p] + cav[t], pb[p] + cbv[t]);
    }
  }
}

struct __attribute__((packed)) unaligned_u32 { uint32_t v; };

// (cost, spiral index, dx, dy) of a key of partition p in column ddx (|ddx| = aa): ordered as JM orders candidates
__device__ __forceinline__ unsigned long long full_key(unsigned kk, int p, int ddx, int aa)
{
  const unsigned cs = cost_shift(p);
  const int ddy = rank_to_dy((kk & ((1u << cs) - 1u)) >> tag_shift(p), aa);
  return ((unsigned long long)(kk >> cs) << 32) | ((unsigned)spiral_index(ddx, ddy) << 16) | ((unsigned)(ddx + 64) << 8) | (unsigned)(ddy + 64);
}

__global__ __launch_bounds__(NT, 4) void k_me_fs_fast(const jmhip_me_job *__restrict__ jobs, jmhip_me_result *__restrict__ results,
                                                       const uint8_t *__restrict__ cur, int cur_pitch,
                                                       const uint8_t *__restrict__ ref00, int pitch, long plane_stride, int W, int H,
                                                       unsigned *__restrict__ declined, int njobs, const unsigned *__restrict__ jerr)
{
  if (*jerr) return;                                        // a job record failed k_check_me_jobs (me_fullsearch.hip)
  __shared__ __attribute__((aligned(16))) uint32_t s_mem[COPY_DWORDS];        // window copies, later the wave-merge area
  __shared__ __attribute__((aligned(16))) uint32_t s_ry[65 * RYP];            // (lambda * mvbits(cand_y - pred_y[p])) << cost_shift, [row][partition]; later the 65th column's keys
  __shared__ uint32_t s_rx64[RYP];                                              // x rate of column 64, per partition
  __shared__ int s_px[RYP];                                                     // x predictor per partition (final stage)
  const int jb = xcd_job_index(blockIdx.x, njobs);
  const jmhip_me_job *__restrict__ job = jobs + jb;
  if (!job_is_fast(job)) {                                  // left to k_me_fullsearch, which only does real work when this counter is non-zero
    if (threadIdx.x == 0) atomicAdd(declined, 1u);
    return;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = job->search_range, n1 = 2 * R + 1, wrows = 2 * R + 16;
  const int cx = job->center_x, cy = job->center_y, lambda = job->lambda;
  const uint64_t mask = job->part_mask;

  // ---- staging: four byte-shifted copies straight from the reference plane of the centre's phase (rows/cols clamped into the
  //      padded plane): copy s, row r, dword k = window bytes 4k+s .. 4k+s+3
  {
    const int x0 = job->mb_x + (cx >> 2) - R, y0 = job->mb_y + (cy >> 2) - R;
    const uint8_t *plane = ref00 + ((cy & 3) * 4 + (cx & 3)) * plane_stride;
    const bool inside = x0 >= -JMHIP_PAD_X && x0 + CPITCH * 4 + 4 <= W + JMHIP_PAD_X;      // no horizontal clamp needed
    // thread <-> dword column c of the copies (20 of them), rows r0, r0 + 12, ...: no division, one address increment per step
    if (tid < 12 * CPITCH) {
      const int r0 = tid / CPITCH, c = tid - r0 * CPITCH;
      const int xa = x0 + 4 * c;
      for (int r = r0; r < wrows; r += 12) {
        const int yy = min(max(y0 + r, -JMHIP_PAD_Y), H + JMHIP_PAD_Y - 1) + JMHIP_PAD_Y;
        const uint8_t *prow = plane + (long)yy * pitch + JMHIP_PAD_X;
        uint32_t d0, d1;
        if (inside) { d0 = ((const unaligned_u32 *)(prow + xa))->v; d1 = ((const unaligned_u32 *)(prow + xa + 4))->v; }
        else {
          d0 = 0; d1 = 0;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            d0 |= (uint32_t)prow[min(max(xa + t, -JMHIP_PAD_X), W + JMHIP_PAD_X - 1)] << (8 * t);
            d1 |= (uint32_t)prow[min(max(xa + 4 + t, -JMHIP_PAD_X), W + JMHIP_PAD_X - 1)] << (8 * t);
          }
        }
        const int k = r * CPITCH + c;
        s_mem[0 * CSTRIDE + k] = d0;
        s_mem[1 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 1);
        s_mem[2 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 2);
        s_mem[3 * CSTRIDE + k] = __builtin_amdgcn_alignbyte(d1, d0, 3);
      }
    }
    // per-partition constants and the y-rate table: thread <-> partition p (6 threads each), rows g, g + 6, ...
    if (tid < 6 * NP) {
      const int g = tid / NP, p = tid - g * NP, cs = cost_shift(p);
      const int py = job->pred[p][1];
      if (g == 0) { const int px = job->pred[p][0]; s_rx64[p] = (uint32_t)__umul24((unsigned)lambda, (unsigned)mvbits(cx + 4 * (64 - R) - px)) << cs; s_px[p] = px; }
      for (int wy = g; wy < n1; wy += 6) s_ry[wy * RYP + p] = (uint32_t)__umul24((unsigned)lambda, (unsigned)mvbits(cy + 4 * (wy - R) - py)) << cs;
    }
  }
  // ---- current macroblock: uniform addresses -> scalar loads, stays in SGPRs
  uint32_t cw[64];
  {
    const uint32_t *cp = (const uint32_t *)(cur + (long)job->mb_y * cur_pitch + job->mb_x);
    const int p4 = cur_pitch >> 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int c = 0; c < 4; c++) cw[r * 4 + c] = cp[r * p4 + c];
    }
  }
  PROF_START;
  __syncthreads();
  PROF(0);

  // ---- main pass: lane = column wx (0..63), wave w takes every NW-th row pair
  const int wx = lane, dx = wx - R, a = dx < 0 ? -dx : dx;
  const bool col_ok = wx < n1;
  unsigned acc[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = 0xffffffffu;
  {
    const int base = (wx & 3) * CSTRIDE + (wx >> 2);
#pragma unroll 1
    for (int u = wave; u <= R; u += NW) step_pair(s_mem, s_ry, base, min(2 * u, 2 * R - 1), R, a, cw, acc);
  }
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = col_ok ? acc[p] : 0xffffffffu;

  PROF(1);
  // ---- 65th column (R = 32): lane j takes rows (2j, 2j+1), j <= R (the last pair overlaps as above); the wave with the fewest
  //      row pairs does it.  Keys get the column's x rate here: they are compared across lanes in the final stage.
  unsigned ex[NP];
  const bool ex_wave = n1 > 64 && wave == NW - 1, ex_live = ex_wave && lane <= R;
  if (ex_wave) {
#pragma unroll
    for (int p = 0; p < NP; p++) ex[p] = 0xffffffffu;
    step_pair(s_mem, s_ry, 64 >> 2, min(2 * lane, 2 * R - 1), R, 64 - R, cw, ex);
#pragma unroll
    for (int p = 0; p < NP; p++) ex[p] += s_rx64[p];
  }
  PROF(2);
  __syncthreads();                                       // every wave is done reading the window copies and the y-rate table
  PROF(3);
  // ---- publish per-lane minima in two rounds (upper half of the waves, then the lower half merged with them):
  //      s_mem[(w*NP + p)*MPITCH + lane], w < NW/2; the 65th column's keys go where the y-rate table was
  uint32_t *s_ext = s_ry;                                // [NP][EXT_PITCH]
  if (wave >= NWH) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_mem[((wave - NWH) * NP + p) * MPITCH + lane] = acc[p];
  }
  if (ex_live) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_ext[p * EXT_PITCH + lane] = ex[p];
  }
  __syncthreads();
  if (wave < NWH) {
#pragma unroll
    for (int p = 0; p < NP; p++) s_mem[(wave * NP + p) * MPITCH + lane] = min(acc[p], s_mem[(wave * NP + p) * MPITCH + lane]);
  }
  __syncthreads();
  PROF(4);
  // ---- final: four threads per partition, each takes 16 columns (+ 9 row pairs of the 65th column).  Within a column the tag
  //      already orders the rows as JM's spiral does; across columns only the cost is comparable and ties go to the lower
  //      spiral index.  So: minimum cost first (32-bit); a thread whose best candidate is alone at that cost turns just that one
  //      into a (cost, spiral index, dx, dy) key; equal costs inside a thread (rare) take the path that converts all of them.
  //      Two xor-shuffles inside the 4-lane group finish it.  No table look-up: the winner carries its own displacement.
  {
    const int p = tid >> 2, sub = tid & 3;
    if (p < NP) {
      const int px = s_px[p], cs = cost_shift(p), dxe = 64 - R;
      unsigned kcol[16], kext[9];
      unsigned bk = 0xffffffffu;                          // minimum over the 32-bit keys: its cost field is this thread's minimal cost
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int col = c * 4 + sub;
        unsigned kk = s_mem[p * MPITCH + col];
#pragma unroll
        for (int w = 1; w < NWH; w++) kk = min(kk, s_mem[(w * NP + p) * MPITCH + col]);
        kk = add_sat(kk, __umul24((unsigned)lambda, (unsigned)mvbits(cx + 4 * (col - R) - px)) << cs);     // "no candidate" stays 0xffffffff
        kcol[c] = kk;
        bk = min(bk, kk);
      }
#pragma unroll
      for (int c = 0; c < 9; c++) {
        const int q = c * 4 + sub;
        unsigned e = 0xffffffffu;
        if (n1 > 64 && q <= R) e = s_ext[p * EXT_PITCH + q];
        kext[c] = e;
        bk = min(bk, e);
      }
      // how many of this thread's candidates share the minimal cost (x ^ bk has no bit at or above cs)
      const unsigned lim = 1u << cs;
      unsigned cnt = 0;
      int bc = 0;
#pragma unroll
      for (int c = 0; c < 16; c++) { cnt += (kcol[c] ^ bk) < lim; bc = kcol[c] == bk ? c : bc; }
#pragma unroll
      for (int c = 0; c < 9; c++) { cnt += (kext[c] ^ bk) < lim; bc = kext[c] == bk ? 16 + c : bc; }
      unsigned long long best = ~0ull;
      if (__builtin_amdgcn_ballot_w64(cnt > 1 && bk != 0xffffffffu) == 0) {
        const int ddx = bc < 16 ? bc * 4 + sub - R : dxe;
        if (bk != 0xffffffffu) best = full_key(bk, p, ddx, ddx < 0 ? -ddx : ddx);
      } else {
        const unsigned m = bk >> cs;
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const int ddx = c * 4 + sub - R;
          if (kcol[c] != 0xffffffffu && (kcol[c] >> cs) == m) { const unsigned long long key = full_key(kcol[c], p, ddx, ddx < 0 ? -ddx : ddx); best = key < best ? key : best; }
        }
#pragma unroll
        for (int c = 0; c < 9; c++)
          if (kext[c] != 0xffffffffu && (kext[c] >> cs) == m) { const unsigned long long key = full_key(kext[c], p, dxe, dxe); best = key < best ? key : best; }
      }
#pragma unroll
      for (int off = 1; off <= 2; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
      }
      if (sub == 0 && ((mask >> p) & 1)) {
        jmhip_me_best b;
        b.mv_x = (int16_t)(cx + 4 * ((int)((best >> 8) & 0xff) - 64)); b.mv_y = (int16_t)(cy + 4 * ((int)(best & 0xff) - 64));
        b.cost = (int32_t)(best >> 32);
        results[jb].best[p] = b;
      }
    }
  }
  PROF(5);
}

void jmhip_launch_me_fast(jmhip_ctx *ctx, int slot, const jmhip_me_job *d_jobs, int njobs, jmhip_me_result *d_results, unsigned *d_declined)
{
  hipLaunchKernelGGL(k_me_fs_fast, dim3(njobs), dim3(NT), 0, ctx->stream, d_jobs, d_results, ctx->d_cur, ctx->cur_pitch,
                     ctx->d_sub[slot], ctx->pitch, (long)ctx->plane_stride, ctx->W, ctx->H, d_declined, njobs, ctx->d_me_declined + 4);
}
