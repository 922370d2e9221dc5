// deblock_sparse.hip -- K9+K10 for pictures in which most macroblocks have nothing to filter: segment walks (gfx950).
//
// Same result, bit for bit, as JM's raster-order DeblockFrame (lencod/src/loopFilter.c:63-297) and as the band pipeline of
// deblock_rows.hip, whose walker this file re-uses (compiled here for one macroblock row per workgroup).
//
// What connects macroblock (x, y) to its left neighbour is its left edge alone: if none of that edge's segments is active (strength 0
// for luma and chroma), nothing the filter does in macroblocks < x of that row is visible to macroblocks >= x.  So a row falls apart at
// those places into independent runs, a run without a single active segment needs no work at all (its samples do not change and the
// hand-over granules of its macroblocks were issued by k_deblock_prep), and the frame's critical path is no longer "every row from the
// left picture edge to the right one" but the longest chain of runs that really depend on each other.  The P picture of
// BASELINE.json configs[1]: 2.8 % of the macroblocks are active, its last (padded) macroblock row is 118 active macroblocks long -- and
// falls into nine runs, the longest 37 macroblocks.
//
//   k_deblock_tasks   one workgroup: from the per-macroblock flags k_deblock_prep left (active / left edge active) the list of runs with
//                     work, in raster order, which macroblocks they cover, and for every covered macroblock whether a run covers the one
//                     below (else the run stores that macroblock's bottom rows itself); decides whether the segment walks do the frame
//                     (few enough tasks and active macroblocks) or the band pipeline does.
//   k_deblock_sparse  one workgroup per (task, plane kind): the walker of deblock_rows.hip on the sub-row [first, end) of its row -- top
//                     rows from the hand-over granules of the row above (pre-issued or written by that row's walks), bottom rows handed
//                     on the same way.  Tasks draw tickets in raster order, so a walk only ever waits for walks that have started.
#define DB_SPARSE 1
#define LR 1
#include "deblock_rows.hip"

#define DB_MAX_TASKS 1024
#define DB_MAX_ROWS 256                            // macroblock rows (and, for the four-word masks, columns) the task builder takes
#define DB_MAX_LDS_MBS 40960                       // macroblocks whose flags it stages in LDS

// Bit masks of one macroblock row (up to 256 macroblocks = four 64-bit words), uniform across the wave that handles the row.
template <int NW_> struct RowMask { unsigned long long w[NW_]; };
// (every loop over the four words is unrolled with the word index as a constant: a run-time index would put the masks in scratch memory,
// and each look at them would be a memory round trip)
template <int NW_> __device__ __forceinline__ int mask_prev_set(const RowMask<NW_> &m, int x)            // highest set bit <= x (bit 0 is always set for a cut mask)
{
  int res = 0;
#pragma unroll
  for (int k = 0; k < NW_; k++) {
    unsigned long long v = m.w[k];
    if (k == (x >> 6)) v &= ~0ull >> (63 - (x & 63)); else if (k > (x >> 6)) v = 0;
    if (v) res = k * 64 + 63 - __clzll((long long)v);
  }
  return res;
}
template <int NW_> __device__ __forceinline__ int mask_next_set(const RowMask<NW_> &m, int x, int n)     // lowest set bit > x, or n
{
  int res = n;
#pragma unroll
  for (int k = NW_ - 1; k >= 0; k--) {
    unsigned long long v = m.w[k];
    if (k == (x >> 6)) v &= (x & 63) == 63 ? 0ull : (~0ull << ((x & 63) + 1)); else if (k < (x >> 6)) v = 0;
    if (v) res = min(n, k * 64 + __ffsll((long long)v) - 1);
  }
  return res;
}
template <int NW_> __device__ __forceinline__ bool mask_any(const RowMask<NW_> &m, int s, int e)         // any set bit in [s, e), s < e
{
  bool r = false;
#pragma unroll
  for (int k = 0; k < NW_; k++) {
    unsigned long long v = m.w[k];
    const int lo = k * 64, hi = lo + 64;
    if (e <= lo || s >= hi) v = 0;
    else { if (s > lo) v &= ~0ull << (s - lo); if (e < hi) v &= ~0ull >> (hi - e); }
    r |= v != 0;
  }
  return r;
}

// One workgroup of sixteen waves; a wave takes every sixteenth row, a lane a macroblock of the row's 64-wide chunks.  Pass 1: per row the mask of macroblocks
// that belong to a run with work (s_proc) and the number of such runs; a scan over the rows gives every row's place in the task list;
// pass 2 writes the tasks and, with the mask of the row below, who stores whose bottom rows.
#define TASK_WAVES 16
template <int NW_>        // 64-bit mask words per row: mb_w <= 64 * NW_
__global__ __launch_bounds__(64 * TASK_WAVES) void k_deblock_tasks(const uint8_t *__restrict__ flags, int mb_w, int mb_h, int2 *__restrict__ tasks,
                                                       uint8_t *__restrict__ store_bottom, unsigned *__restrict__ ctl, int max_active_pct)
{
  __shared__ unsigned long long s_proc[DB_MAX_ROWS + 1][NW_];
  __shared__ int s_cnt[DB_MAX_ROWS], s_off[DB_MAX_ROWS], s_act[DB_MAX_ROWS], s_wsum[4], s_asum[4];
  // the flags go through LDS: every lane fetches its share with a few wide loads that are all in flight together (the buffer is padded
  // to a multiple of 16), so the kernel is one memory latency long however many rows a wave takes
  __shared__ __attribute__((aligned(16))) uint8_t s_f[DB_MAX_LDS_MBS + 16];
  {
    const int nvec = (mb_w * mb_h + 15) >> 4, i0 = threadIdx.x, i1 = i0 + 64 * TASK_WAVES, i2 = i1 + 64 * TASK_WAVES;   // 3 x 1024 x 16 >= DB_MAX_LDS_MBS
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 v0 = i0 < nvec ? ((const uint4 *)flags)[i0] : z, v1 = i1 < nvec ? ((const uint4 *)flags)[i1] : z, v2 = i2 < nvec ? ((const uint4 *)flags)[i2] : z;
    if (i0 < nvec) ((uint4 *)s_f)[i0] = v0;
    if (i1 < nvec) ((uint4 *)s_f)[i1] = v1;
    if (i2 < nvec) ((uint4 *)s_f)[i2] = v2;
  }
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nmb = mb_w * mb_h, nch = (mb_w + 63) >> 6;
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
    for (int r = wave; r < mb_h; r += TASK_WAVES) {
      RowMask<NW_> cut, act;
#pragma unroll
      for (int c = 0; c < NW_; c++) {                                            // (no early exit: the word index must stay a constant)
        const int x = c * 64 + lane;
        const int f = x < mb_w ? s_f[r * mb_w + x] : 0;
        cut.w[c] = __ballot(x < mb_w && (x == 0 || !(f & 2)));                  // a cut before x: nothing connects x to x - 1
        act.w[c] = __ballot(f & 1);
      }
      int cnt = 0, nact = 0, base = pass ? s_off[r] : 0;
#pragma unroll
      for (int c = 0; c < NW_; c++) {
        const int x = c * 64 + lane;
        bool proc = false, start = false;
        int e = 0;
        if (x < mb_w) {
          const int sx = mask_prev_set(cut, x);
          e = mask_next_set(cut, x, mb_w);
          proc = mask_any(act, sx, e);
          start = proc && sx == x;
        }
        const unsigned long long pm = __ballot(proc), sm = __ballot(start);
        if (pass == 0) { if (lane == 0 && c < nch) s_proc[r][c] = pm; }
        else {
          if (start) tasks[base + cnt + __popcll(sm & ((1ull << lane) - 1))] = make_int2(r, x | (e << 16));
          if (x < mb_w) store_bottom[r * mb_w + x] = (uint8_t)(proc && r + 1 < mb_h && !((s_proc[r + 1][c] >> lane) & 1));
        }
        cnt += __popcll(sm); nact += __popcll(act.w[c]);
      }
      if (pass == 0 && lane == 0) { s_cnt[r] = cnt; s_act[r] = nact; }
    }
    if (pass == 1) break;
    __syncthreads();
    // exclusive scan of the rows' task counts (and the total of active macroblocks): rows <= 256 = one per thread
    int v = tid < mb_h ? s_cnt[tid] : 0, a = tid < mb_h ? s_act[tid] : 0, inc = v;          // rows <= 256: waves 0..3 hold them
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 63 && wave < 4) s_wsum[wave] = inc;
    if (lane == 0 && wave < 4) s_asum[wave] = a;
    __syncthreads();
    int woff = 0, total = 0, active = 0;
    for (int k = 0; k < 4; k++) { if (k < wave) woff += s_wsum[k]; total += s_wsum[k]; active += s_asum[k]; }
    if (tid < mb_h) s_off[tid] = woff + inc - v;
    const bool sparse = total <= DB_MAX_TASKS && active * 100 <= nmb * max_active_pct;
    if (tid == 0) { ctl[1] = (unsigned)total; ctl[0] = sparse ? 1u : 0u; }
    if (!sparse) return;                                                          // workgroup-uniform
    __syncthreads();
  }
}

__global__ __launch_bounds__(192) void k_deblock_sparse(RowArgs A)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_tiles[4 * (YT_BYTES > CT_BYTES ? YT_BYTES : CT_BYTES)];
  __shared__ __attribute__((aligned(16))) uint8_t s_preps[4 * 2 * sizeof(DbPrep)];
  __shared__ unsigned s_ticket;
  __shared__ int s_abort;
  if (A.ctl[0] != 1u) return;                       // the band pipeline does this frame
  // the grid is sized for the largest task list; only as many workgroups as there are (task, kind) pairs draw a ticket (2048 atomics on
  // one word would take longer than the walks)
  if (blockIdx.x >= A.ctl[1] * (unsigned)A.nkinds) return;
  const int tid = threadIdx.x;
  if (tid == 0) { s_ticket = __hip_atomic_fetch_add((gu32 *)(A.ctl + 2), 1u, RLX_AGENT); s_abort = 0; }
  for (int k = tid; k < (int)sizeof(s_tiles) / 4; k += 192) ((uint32_t *)s_tiles)[k] = 0;
  __syncthreads();
  const int t = (int)s_ticket, task = t / A.nkinds, kind = t - task * A.nkinds;
  if (task >= (int)A.ctl[1]) return;
  const int2 tk = A.tasks[task];
  const int row = tk.x, x0 = tk.y & 0xffff, x1 = tk.y >> 16;
  RowArgs T = A;
  T.Y += 16 * x0; T.prep += x0; T.hand += (long)x0 * HAND_PER_MB; T.store_bottom += x0; T.mb_w = x1 - x0;
  if (A.nkinds > 1) { T.U += 8 * x0; T.V += 8 * x0; }
  if (kind == 0) luma_rows(T, row, s_tiles, s_preps, (lds_int *)&s_abort);
  else chroma_rows(T, row, s_tiles, s_preps, (lds_int *)&s_abort);
}

// after k_deblock_prep on the context's stream; A as the band pipeline gets it (stride, ctl, tasks, store_bottom set by the caller)
int jmhip_launch_deblock_sparse(jmhip_ctx *ctx, const RowArgs &A, const uint8_t *d_flags, int max_active_pct)
{
  if (A.mb_w <= 64) hipLaunchKernelGGL(k_deblock_tasks<1>, dim3(1), dim3(64 * TASK_WAVES), 0, ctx->stream, d_flags, A.mb_w, A.mb_h, (int2 *)A.tasks, (uint8_t *)A.store_bottom, A.ctl, max_active_pct);
  else if (A.mb_w <= 128) hipLaunchKernelGGL(k_deblock_tasks<2>, dim3(1), dim3(64 * TASK_WAVES), 0, ctx->stream, d_flags, A.mb_w, A.mb_h, (int2 *)A.tasks, (uint8_t *)A.store_bottom, A.ctl, max_active_pct);
  else hipLaunchKernelGGL(k_deblock_tasks<4>, dim3(1), dim3(64 * TASK_WAVES), 0, ctx->stream, d_flags, A.mb_w, A.mb_h, (int2 *)A.tasks, (uint8_t *)A.store_bottom, A.ctl, max_active_pct);
  hipLaunchKernelGGL(k_deblock_sparse, dim3(DB_MAX_TASKS * A.nkinds), dim3(192), 0, ctx->stream, A);
  return JMHIP_OK;
}
