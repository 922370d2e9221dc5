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

// ---------------------------------------------------------------------------------------------------
// Column walks.  A run of ONE macroblock has no horizontal neighbour to wait for or to hand anything to (its left edge and the left
// edge of its right neighbour are inactive), so its vertical edges V1..V3 need nothing but its own samples, and what chains such
// macroblocks when they sit on top of each other (a picture border with new content in every row) is the horizontal edges alone:
// H0..H3 of the macroblock, then H0 of the one below across the rows H3 just left.  One workgroup takes up to DB_COL_ROWS of them:
//   every thread loads one sample row (and its part of the strength records), filters the row's vertical edges in registers -- all rows
//   of all macroblocks at once -- and puts it into a tall LDS tile;
//   sixteen lanes (lane = sample column) then walk down the tile: four dependent edge filters per macroblock, the bottom rows staying in
//   registers as the top rows of the next macroblock -- no hand-over through memory inside the chain;
//   granules for the macroblock below the chain, image stores of everything (same ownership as the row walks: the top neighbour's
//   bottom rows are stored here, the chain's own bottom rows only where no walk covers the macroblock below).
// The top neighbour's granules are polled from the start and waited for only before the horizontal edges.
#define DB_COL_ROWS 16
#define YC_BYTES ((4 + 16 * DB_COL_ROWS) * 16)
#define CC_PLANE ((2 + 16 * DB_COL_ROWS) * 8)

__device__ void luma_col(const RowArgs &A, int r0, int n, uint8_t *s_tile, uint8_t *s_preps, volatile lds_int *s_abort)
{
  const int tid = threadIdx.x, lane = tid & 63;
  gu32 *err = (gu32 *)(A.sync + 1);
  const bool has_up = r0 > 0, has_down = r0 + n < A.mb_h;
  uint8_t *band_p = A.Y + (long)(16 * r0) * A.pitchY;
  const int nr = 16 * n;
  const bool is_gran = has_up && tid >= 128 && lane < 16;                      // granule `lane` = top row lane >> 2, dword lane & 3
  const unsigned long long *hand_up = A.hand + (long)(r0 - 1) * A.stride * HAND_PER_MB + lane;
  unsigned long long gr = 0;
  if (is_gran) gr = get_granule(hand_up);
  // ---- own rows (two per thread at most), their vertical-edge parameters straight from the records, the records for the walk into LDS
  const int R0 = tid, R1 = tid + 192;
  const uint4 z = make_uint4(0, 0, 0, 0);
  uint4 v0 = z, v1 = z;
  uint32_t bs0 = 0, bs1 = 0, c00 = 0, c01 = 0, ab0 = 0, ab1 = 0;
  if (R0 < nr) {
    const DbPrep *P = A.prep + (long)(r0 + (R0 >> 4)) * A.stride;
    v0 = *(const uint4 *)(band_p + (long)R0 * A.pitchY);
    bs0 = *(const uint32_t *)&P->bsY[0][(R0 & 15) >> 2][0]; c00 = *(const uint32_t *)&P->c0Y[0][(R0 & 15) >> 2][0]; ab0 = *(const uint16_t *)&P->ab[0][2][0];
  }
  if (R1 < nr) {
    const DbPrep *P = A.prep + (long)(r0 + (R1 >> 4)) * A.stride;
    v1 = *(const uint4 *)(band_p + (long)R1 * A.pitchY);
    bs1 = *(const uint32_t *)&P->bsY[0][(R1 & 15) >> 2][0]; c01 = *(const uint32_t *)&P->c0Y[0][(R1 & 15) >> 2][0]; ab1 = *(const uint16_t *)&P->ab[0][2][0];
  }
  if (tid < n * PREP_VEC) { const int pg = tid / PREP_VEC, pv = tid - pg * PREP_VEC; ((uint4 *)(s_preps + pg * sizeof(DbPrep)))[pv] = ((const uint4 *)(A.prep + (long)(r0 + pg) * A.stride))[pv]; }
  // ---- V1..V3 (the left edge of a one-macroblock run is inactive): columns 0..15 in registers
#define VROW(v, bsv, c0v, abv) { \
    const uint32_t bs = (bsv) & 0xffffff00u; \
    if (bs) { \
      const uint32_t w[4] = {v.x, v.y, v.z, v.w}; \
      int p[20]; \
      p[0] = p[1] = p[2] = p[3] = 0; \
      _Pragma("unroll") for (int k = 0; k < 16; k++) p[4 + k] = (w[k >> 2] >> (8 * (k & 3))) & 255; \
      VEDGE(1, bs, c0v, abv) VEDGE(2, bs, c0v, abv) VEDGE(3, bs, c0v, abv) \
      v = make_uint4(pack4(p[4], p[5], p[6], p[7]), pack4(p[8], p[9], p[10], p[11]), pack4(p[12], p[13], p[14], p[15]), pack4(p[16], p[17], p[18], p[19])); \
    } }
#define VEDGE(E, bs, c0v, abv) { const int bS = (bs >> (8 * (E))) & 255; \
                   if (__any(bS)) luma_edge<4 * (E), 20>(p, bS, (abv) & 255, (abv) >> 8, ((c0v) >> (8 * (E))) & 255, __any(bS == 4)); }
  VROW(v0, bs0, c00, ab0)
  VROW(v1, bs1, c01, ab1)
#undef VEDGE
#undef VROW
  if (R0 < nr) *(uint4 *)(s_tile + (4 + R0) * 16) = v0;
  if (R1 < nr) *(uint4 *)(s_tile + (4 + R1) * 16) = v1;
  if (tid >= 128 && has_up) {                                                  // wave-uniform
    if (!__all(!is_gran || (gr >> 32) != 0)) { if (!await_granules(hand_up, is_gran, gr, err)) *s_abort = 1; }
    if (is_gran) ((uint32_t *)(s_tile + (lane >> 2) * 16))[lane & 3] = (uint32_t)gr;
  }
  __syncthreads();
  if (*s_abort) return;
  // ---- the walk down: lane = sample column; p[0..3] the rows above the macroblock, p[4..19] its own
  if (tid < 16) {
    const int l = tid, seg = l >> 2;
    int p[20], q[16];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = s_tile[k * 16 + l];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = s_tile[(4 + k) * 16 + l];
#pragma unroll 1
    for (int m = 0; m < n; m++) {
      uint8_t *c = s_tile + (16 * m) * 16 + l;
      const DbPrep *P = (const DbPrep *)(s_preps + m * sizeof(DbPrep));
      const uint32_t bs = *(const uint32_t *)&P->bsY[1][seg][0];
      const uint32_t c0 = *(const uint32_t *)&P->c0Y[1][seg][0];
      const uint32_t abE = *(const uint16_t *)&P->ab[0][1][0], abI = *(const uint16_t *)&P->ab[0][2][0];
#pragma unroll
      for (int k = 0; k < 16; k++) p[4 + k] = q[k];
      if (m + 1 < n) {                                                         // the next macroblock's rows, before the filters
#pragma unroll
        for (int k = 0; k < 16; k++) q[k] = c[(20 + k) * 16];
      }
      if (bs) {
#define HEDGE(E) { const int bS = (bs >> (8 * (E))) & 255; const uint32_t ab = (E) ? abI : abE; \
                   if (__any(bS)) luma_edge<4 * (E), 20>(p, bS, ab & 255, ab >> 8, (c0 >> (8 * (E))) & 255, __any(bS == 4)); }
        HEDGE(0) HEDGE(1) HEDGE(2) HEDGE(3)
#undef HEDGE
#pragma unroll
        for (int k = 1; k < 19; k++) c[k * 16] = (uint8_t)p[k];
      }
      p[0] = p[16]; p[1] = p[17]; p[2] = p[18]; p[3] = p[19];
    }
  }
  __syncthreads();
  // ---- hand-over of the last macroblock's bottom rows, image stores
  if (has_down && tid >= 64 && tid < 80) {
    const int g = tid - 64;
    put_granule(A.hand + (long)(r0 + n - 1) * A.stride * HAND_PER_MB + g, ((const uint32_t *)(s_tile + (nr + (g >> 2)) * 16))[g & 3]);
  }
  const bool tail = !has_down || A.store_bottom[(long)(r0 + n - 1) * A.stride];
  const int T0 = has_up ? 0 : 4, T1 = nr + (tail ? 4 : 0);                     // tile row T = picture row 16 r0 - 4 + T
  for (int T = tid; T < T1; T += 192)
    if (T >= T0) *(uint4 *)(band_p + (long)(T - 4) * A.pitchY) = *(const uint4 *)(s_tile + T * 16);
}

__device__ void chroma_col(const RowArgs &A, int r0, int n, uint8_t *s_tile /* 2 x CC_PLANE */, uint8_t *s_preps, volatile lds_int *s_abort)
{
  const int tid = threadIdx.x, lane = tid & 63, fmt = A.fmt, RH = fmt == 2 ? 16 : 8;
  gu32 *err = (gu32 *)(A.sync + 1);
  const bool has_up = r0 > 0, has_down = r0 + n < A.mb_h;
  const int nr = RH * n;                                                       // sample rows per plane
  const int gq = lane & 7;                                                     // granule: plane (bit 2), top row (bit 1), dword (bit 0)
  const bool is_gran = has_up && tid >= 128 && lane < 8;
  const unsigned long long *hand_up = A.hand + (long)(r0 - 1) * A.stride * HAND_PER_MB + 16 + gq;
  unsigned long long gr = 0;
  if (is_gran) gr = get_granule(hand_up);
  // ---- own rows of both planes (three per thread at most): row i = plane i / nr, sample row i % nr
  uint2 v[3];
  uint32_t bsv[3], c0v[3], abv[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int i = tid + 192 * j;
    v[j] = make_uint2(0, 0); bsv[j] = 0; c0v[j] = 0; abv[j] = 0;
    if (i < 2 * nr) {
      const int uv = i >= nr, R = i - uv * nr, m = R / RH, rr = R - m * RH, seg = RH == 8 ? (rr >> 1) : (rr >> 2);
      const DbPrep *P = A.prep + (long)(r0 + m) * A.stride;
      v[j] = *(const uint2 *)((uv ? A.V : A.U) + (long)(RH * r0 + R) * A.pitchC);
      bsv[j] = *(const uint32_t *)&P->bsC[0][seg][0]; c0v[j] = *(const uint32_t *)&P->c0C[uv][0][seg][0]; abv[j] = *(const uint16_t *)&P->ab[1 + uv][2][0];
    }
  }
  if (tid < n * PREP_VEC) { const int pg = tid / PREP_VEC, pv = tid - pg * PREP_VEC; ((uint4 *)(s_preps + pg * sizeof(DbPrep)))[pv] = ((const uint4 *)(A.prep + (long)(r0 + pg) * A.stride))[pv]; }
  // ---- the vertical edge inside the macroblock (luma edge 2 -> chroma column 4); the left edge is inactive
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int i = tid + 192 * j;
    const int bS = (bsv[j] >> 16) & 255;
    if (bS) {
      const uint32_t w[2] = {v[j].x, v[j].y};
      int p[8];
#pragma unroll
      for (int k = 0; k < 8; k++) p[k] = (w[k >> 2] >> (8 * (k & 3))) & 255;
      chroma_edge4(p[2], p[3], p[4], p[5], bS, abv[j] & 255, abv[j] >> 8, (c0v[j] >> 16) & 255);
      v[j] = make_uint2(pack4(p[0], p[1], p[2], p[3]), pack4(p[4], p[5], p[6], p[7]));
    }
    if (i < 2 * nr) { const int uv = i >= nr, R = i - uv * nr; *(uint2 *)(s_tile + uv * CC_PLANE + (2 + R) * 8) = v[j]; }
  }
  if (tid >= 128 && has_up) {
    if (!__all(!is_gran || (gr >> 32) != 0)) { if (!await_granules(hand_up, is_gran, gr, err)) *s_abort = 1; }
    if (is_gran) *(uint32_t *)(s_tile + ((gq >> 2) & 1) * CC_PLANE + ((gq >> 1) & 1) * 8 + (gq & 1) * 4) = (uint32_t)gr;
  }
  __syncthreads();
  if (*s_abort) return;
  // ---- the walk down: lane = (plane, column); p[0..1] the rows above the macroblock, p[2..RH+1] its own
  if (tid < 16) {
    const int uvc = tid >> 3, cc = tid & 7, seg = cc >> 1;
    uint8_t *c0p = s_tile + uvc * CC_PLANE + cc;
    int p[18], q[16];
    p[0] = c0p[0]; p[1] = c0p[8];
#pragma unroll
    for (int k = 0; k < 16; k++) { q[k] = (k < RH) ? c0p[(2 + k) * 8] : 0; p[2 + k] = 0; }
#pragma unroll 1
    for (int m = 0; m < n; m++) {
      uint8_t *c = c0p + (RH * m) * 8;
      const DbPrep *P = (const DbPrep *)(s_preps + m * sizeof(DbPrep));
      const uint32_t bs = *(const uint32_t *)&P->bsC[1][seg][0];
      const uint32_t c0 = *(const uint32_t *)&P->c0C[uvc][1][seg][0];
      const uint32_t abE = *(const uint16_t *)&P->ab[1 + uvc][1][0], abI = *(const uint16_t *)&P->ab[1 + uvc][2][0];
#pragma unroll
      for (int k = 0; k < 16; k++) p[2 + k] = q[k];
      if (m + 1 < n) {
#pragma unroll
        for (int k = 0; k < 16; k++) if (k < RH) q[k] = c[(RH + 2 + k) * 8];
      }
      if (bs) {
        // chroma_edge[1][e][fmt]: 4:2:0 -> rows 0 (e = 0), 4 (e = 2); 4:2:2 -> rows 0, 4, 8, 12 (e = 0..3)
#define CHEDGE(E, ROW) { const uint32_t ab = (E) ? abI : abE; \
                         chroma_edge4(p[ROW], p[(ROW) + 1], p[(ROW) + 2], p[(ROW) + 3], (bs >> (8 * (E))) & 255, ab & 255, ab >> 8, (c0 >> (8 * (E))) & 255); }
        CHEDGE(0, 0)
        if (fmt == 1) { CHEDGE(2, 4) }
        else { CHEDGE(1, 4) CHEDGE(2, 8) CHEDGE(3, 12) }
#undef CHEDGE
#pragma unroll
        for (int k = 1; k < 17; k++) if (k < RH + 1) c[k * 8] = (uint8_t)p[k];
      }
      if (RH == 8) { p[0] = p[8]; p[1] = p[9]; } else { p[0] = p[16]; p[1] = p[17]; }
    }
  }
  __syncthreads();
  if (has_down && tid >= 64 && tid < 72) {
    const int q = tid - 64, uv = q >> 2, r = (q >> 1) & 1, c4 = q & 1;
    put_granule(A.hand + (long)(r0 + n - 1) * A.stride * HAND_PER_MB + 16 + q, *((const uint32_t *)(s_tile + uv * CC_PLANE + (nr + r) * 8) + c4));
  }
  const bool tail = !has_down || A.store_bottom[(long)(r0 + n - 1) * A.stride];
  const int T0 = has_up ? 0 : 2, T1 = nr + (tail ? 2 : 0);                     // tile row T = picture row RH r0 - 2 + T
  for (int i = tid; i < 2 * T1; i += 192) {
    const int uv = i >= T1, T = i - uv * T1;
    if (T >= T0) *(uint2 *)((uv ? A.V : A.U) + (long)(RH * r0 - 2 + T) * A.pitchC) = *(const uint2 *)(s_tile + uv * CC_PLANE + T * 8);
  }
}

#define DB_MAX_TASKS 1024
#define DB_MAX_ROWS 256                            // macroblock rows (and, for the four-word masks, columns) the task builder takes

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

// One single-wave workgroup per macroblock row (a lane takes a macroblock of each of the row's 64-wide chunks), so the rows are worked on by
// as many compute units at once: (1) the row's masks -- macroblocks that belong to a run with work, runs of one macroblock -- and its
// number of runs and of active macroblocks, published as {tag, active, runs} in ONE release store after the masks (the data is the flag,
// as with the hand-over granules; k_deblock_prep zeroes the words); (2) every workgroup reads all rows' words (at most 256: four per
// lane, bounded spin -- at most 256 one-wave workgroups are always resident together), which gives it its place in the task list, the
// totals and thus the frame's mode; (3) it writes its tasks and, with the masks of the row below / of its group of DB_COL_ROWS rows, who
// stores whose bottom rows and how long the column walks are.
#define DB_MASK_WORDS 8                            // per row in the mask scratch: [0,4) runs with work, [4,8) runs of one macroblock
template <int NW_>        // 64-bit mask words per row: mb_w <= 64 * NW_
__global__ __launch_bounds__(64) void k_deblock_tasks(const uint8_t *__restrict__ flags, int mb_w, int mb_h, int2 *__restrict__ tasks,
                                                      uint8_t *__restrict__ store_bottom, unsigned *__restrict__ ctl, int max_active_pct,
                                                      unsigned long long *rowinfo, unsigned long long *masks, unsigned *err)
{
  __shared__ unsigned long long s_one[DB_COL_ROWS][NW_];
  const int lane = threadIdx.x, r = blockIdx.x, nmb = mb_w * mb_h;
  RowMask<NW_> cut, act;
#pragma unroll
  for (int c = 0; c < NW_; c++) {                                              // (no early exit: the word index must stay a constant)
    const int x = c * 64 + lane;
    const int f = x < mb_w ? flags[r * mb_w + x] : 0;
    cut.w[c] = __ballot(x < mb_w && (x == 0 || !(f & 2)));                    // a cut before x: nothing connects x to x - 1
    act.w[c] = __ballot(f & 1);
  }
  int cnt = 0, nact = 0, e[NW_];
  bool proc[NW_], start[NW_];
  unsigned long long sm[NW_];
#pragma unroll
  for (int c = 0; c < NW_; c++) {
    const int x = c * 64 + lane;
    proc[c] = false; start[c] = false; e[c] = 0;
    if (x < mb_w) {
      const int sx = mask_prev_set(cut, x);
      e[c] = mask_next_set(cut, x, mb_w);
      proc[c] = mask_any(act, sx, e[c]);
      start[c] = proc[c] && sx == x;
    }
    const unsigned long long pm = __ballot(proc[c]), om = __ballot(start[c] && e[c] == x + 1);
    sm[c] = __ballot(start[c]);
    if (lane == 0) {
      __hip_atomic_store((gu64 *)(masks + (long)r * DB_MASK_WORDS + c), pm, RLX_AGENT);
      __hip_atomic_store((gu64 *)(masks + (long)r * DB_MASK_WORDS + 4 + c), om, RLX_AGENT);
    }
    cnt += __popcll(sm[c]); nact += __popcll(act.w[c]);
  }
  if (lane == 0) __hip_atomic_store((gu64 *)(rowinfo + r), (1ull << 63) | ((unsigned long long)nact << 32) | (unsigned)cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // ---- all rows' words
  int before = 0, total = 0, active = 0;
  {
    unsigned long long v[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) if (k * 64 + lane < mb_h) v[k] = __hip_atomic_load((gu64 *)(rowinfo + k * 64 + lane), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 0;; spins++) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 4; k++) ok = ok && (k * 64 + lane >= mb_h || (v[k] >> 63) != 0);
      if (__all(ok)) break;
      if (spins > DB_SPIN_LIMIT || ((spins & 31u) == 31u && __hip_atomic_load((gu32 *)err, RLX_AGENT) != 0)) { __hip_atomic_store((gu32 *)err, 1u, RLX_AGENT); return; }
#pragma unroll
      for (int k = 0; k < 4; k++) if (k * 64 + lane < mb_h && (v[k] >> 63) == 0) v[k] = __hip_atomic_load((gu64 *)(rowinfo + k * 64 + lane), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)(v[k] & 0xffffffffu), a = (int)((v[k] >> 32) & 0x7fffffffu);
      total += c; active += a; if (k * 64 + lane < r) before += c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { total += __shfl_xor(total, d, 64); active += __shfl_xor(active, d, 64); before += __shfl_xor(before, d, 64); }
  }
  const bool sparse = total <= DB_MAX_TASKS && active * 100 <= nmb * max_active_pct;
  if (r == 0 && lane == 0) { ctl[1] = (unsigned)total; ctl[0] = sparse ? 1u : 0u; }
  if (!sparse) return;                                                          // the same for every workgroup
  // ---- the tasks of this row
  const int g0 = r - (r % DB_COL_ROWS);                                         // its group of rows for the column walks
  if (lane < DB_COL_ROWS * NW_) {
    const int rr = g0 + lane / NW_, c = lane % NW_;
    s_one[lane / NW_][c] = rr < mb_h ? __hip_atomic_load((gu64 *)(masks + (long)rr * DB_MASK_WORDS + 4 + c), RLX_AGENT) : 0ull;
  }
  unsigned long long below[NW_];
#pragma unroll
  for (int c = 0; c < NW_; c++) below[c] = r + 1 < mb_h ? __hip_atomic_load((gu64 *)(masks + (long)(r + 1) * DB_MASK_WORDS + c), RLX_AGENT) : 0ull;
  __syncthreads();
  int base = before;
#pragma unroll
  for (int c = 0; c < NW_; c++) {
    const int x = c * 64 + lane;
    // a run of one macroblock right below another one (same column, same group of DB_COL_ROWS rows) is filtered by the column walk that
    // starts at the top of that chain: its task stays in the list with length 0
    int len = 1;
    if (start[c] && e[c] == x + 1) {
      if (r > g0 && ((s_one[r - g0 - 1][c] >> lane) & 1)) len = 0;
      else for (int rr = r + 1; rr < mb_h && rr < g0 + DB_COL_ROWS && ((s_one[rr - g0][c] >> lane) & 1); rr++) len++;
    }
    if (start[c]) tasks[base + __popcll(sm[c] & ((1ull << lane) - 1))] = make_int2(r | (len << 16), x | (e[c] << 16));
    if (x < mb_w) store_bottom[r * mb_w + x] = (uint8_t)(proc[c] && r + 1 < mb_h && !((below[c] >> lane) & 1));
    base += __popcll(sm[c]);
  }
}

__global__ __launch_bounds__(192) void k_deblock_sparse(RowArgs A)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_tiles[4 * (YT_BYTES > CT_BYTES ? YT_BYTES : CT_BYTES)];
  __shared__ __attribute__((aligned(16))) uint8_t s_preps[4 * 2 * sizeof(DbPrep)];
  __shared__ __attribute__((aligned(16))) uint8_t s_col[YC_BYTES > 2 * CC_PLANE ? YC_BYTES : 2 * CC_PLANE];
  __shared__ __attribute__((aligned(16))) uint8_t s_cpre[DB_COL_ROWS * sizeof(DbPrep)];
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
  const int row = tk.x & 0xffff, len = tk.x >> 16, x0 = tk.y & 0xffff, x1 = tk.y >> 16;
  if (len == 0) return;                             // part of the column walk that starts above
  RowArgs T = A;
  T.Y += 16 * x0; T.prep += x0; T.hand += (long)x0 * HAND_PER_MB; T.store_bottom += x0; T.mb_w = x1 - x0;
  if (A.nkinds > 1) { T.U += 8 * x0; T.V += 8 * x0; }
  if (x1 - x0 == 1) {
    if (kind == 0) luma_col(T, row, len, s_col, s_cpre, (lds_int *)&s_abort);
    else chroma_col(T, row, len, s_col, s_cpre, (lds_int *)&s_abort);
  }
  else if (kind == 0) luma_rows(T, row, s_tiles, s_preps, (lds_int *)&s_abort);
  else chroma_rows(T, row, s_tiles, s_preps, (lds_int *)&s_abort);
}

// after k_deblock_prep on the context's stream; A as the band pipeline gets it (stride, ctl, tasks, store_bottom set by the caller)
int jmhip_launch_deblock_sparse(jmhip_ctx *ctx, const RowArgs &A, const uint8_t *d_flags, int max_active_pct)
{
  unsigned long long *rowinfo = (unsigned long long *)(A.sync + DB_SYNC_ROWINFO), *masks = (unsigned long long *)((char *)ctx->d_db_tasks + DB_MAX_TASKS * 8);
#define TASKS(NW) hipLaunchKernelGGL(k_deblock_tasks<NW>, dim3(A.mb_h), dim3(64), 0, ctx->stream, d_flags, A.mb_w, A.mb_h, (int2 *)A.tasks, (uint8_t *)A.store_bottom, A.ctl, max_active_pct, rowinfo, masks, A.sync + 1)
  if (A.mb_w <= 64) TASKS(1); else if (A.mb_w <= 128) TASKS(2); else TASKS(4);
#undef TASKS
  hipLaunchKernelGGL(k_deblock_sparse, dim3(DB_MAX_TASKS * A.nkinds), dim3(192), 0, ctx->stream, A);
  return JMHIP_OK;
}
