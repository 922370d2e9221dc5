// me_refine.hip -- K4 for whole macroblocks: BlockMotionSearch's IntPelME -> SubPelME hand-over (lencod/src/mv_search.c:960-981)
// for every searched partition of every window job, device resident (gfx950).
//
// Same result, bit for bit, as sub_pel_motion_estimation (lencod/src/me_fullsearch.c:186-289, RDOptimization != 0) applied to
// each partition, and as the per-block kernel k_me_subpel (me_subpel.hip), with the distortions of
//   computeSAD              me_distortion.c:349-426   (block origin through UMVLine4X, refbuf.h:22-26)
//   computeSATD             me_distortion.c:745-825   (per 4x4 -- or per 8x8 when test8x8 -- sub-block origin through UMVLine4X)
//   HadamardSAD4x4 / 8x8    me_distortion.c:175-258 / :266-341
//
// Mapping.  One workgroup (two waves) = one window job = one macroblock.  Whatever the block type, a macroblock is sixteen 4x4 blocks, so
// one refinement stage (9 candidates) of all 7 block types is 7 x 16 x 9 = 1008 equal work items "SAD / Hadamard-SATD of 4x4 block
// b at candidate c of the partition of type t that contains b"; a lane owns one of the 112 (t, b) pairs and walks its nine candidates,
// with no divergence between block sizes.  A candidate reads its four reference rows as (unaligned) dwords straight from the
// plane its quarter-pel phase selects, the current macroblock sits in LDS, and the per-(partition, candidate)
// sums are formed with LDS atomics.  41 lanes then replay JM's sequential strict-'<' scan over the 9 costs.
// With the 8x8 transform (test8x8, block types 1-4) the unit is an 8x8 block (Hadamard 8x8), led by the lane of its top-left 4x4.
#include "jmhip_internal.h"
#include "me_common.h"

struct PlaneSet2 { const uint8_t *base; int pitch; long plane_stride; int W, H; const unsigned *jerr; };   // jerr: the context's job error word (me_fullsearch.hip)

__device__ __forceinline__ int iabs2_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int mvbits2(int d) { int a = iabs2_(d); return a == 0 ? 1 : 2 * (31 - __clz(a)) + 3; }
// Byte offset, from the first plane, of the sample UMVLine4X (refbuf.h:22-26) yields for quarter-pel position (qx, qy): the plane the two low
// bits select, the ORIGIN clamped.  32-bit throughout (16 planes of an 8K picture are 0.5 GB) and with 24-bit multiplies -- v_mul_u32_u24 runs
// at full rate, a 64-bit multiply-add at a quarter of it, and with a uniform base the loads take the offset as it is (saddr addressing), so
// no 64-bit address arithmetic is left in the kernel.  plane_stride is a multiple of 256 (jmhip_create), pitch and row numbers are < 2^24.
__device__ __forceinline__ uint32_t umv_off2(const PlaneSet2 &ps, int qy, int qx)
{
  const int yy = min(max(qy >> 2, -JMHIP_PAD_Y), ps.H + 3), xx = min(max(qx >> 2, -JMHIP_PAD_X), ps.W + 15);
  return (__umul24((uint32_t)((qy & 3) * 4 + (qx & 3)), (uint32_t)(ps.plane_stride >> 8)) << 8) + __umul24((uint32_t)(yy + JMHIP_PAD_Y), (uint32_t)ps.pitch) +
         (uint32_t)(xx + JMHIP_PAD_X);
}
struct __attribute__((packed)) u32u { uint32_t v; };
__device__ __forceinline__ uint32_t ld4(const uint8_t *p) { return ((const u32u *)p)->v; }
__device__ __forceinline__ uint32_t ld4o(const uint8_t *base, uint32_t off) { return ((const u32u *)(base + (size_t)off))->v; }

// (sum |H d H^T| + 1) >> 1 of a 4x4 block given as four packed rows of the current and the reference samples: HadamardSAD4x4.
// Packed 16-bit arithmetic, two sample rows per instruction (every intermediate fits: |row pass| <= 1020, |column pass| <= 4080):
//   v_perm_b32 lifts column k of two rows into the halves of a register (4 + 4 per row pair), v_pk_sub_i16 forms the differences,
//   the row butterflies are 8 v_pk_add/sub per row pair; in the column pass P = rows (0,1), Q = rows (2,3) of one column give
//   [s0, s1] = P + swap(Q) and [s3, s2] = P - swap(Q), and since |x + y| + |x - y| = 2 max(|x|, |y|) the four outputs of the column
//   add up to 2 max(|s0|, |s1|) + 2 max(|s2|, |s3|): no last butterfly, and the final (sum + 1) >> 1 is just the sum of the maxima.
// About half the instructions of the scalar form (this kernel is bound by VALU issue).
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v colpair(uint32_t row_lo, uint32_t row_hi, int k)
{
  return __builtin_bit_cast(s2v, __builtin_amdgcn_perm(row_hi, row_lo, 0x0c040c00u + (uint32_t)k * 0x00010001u));   // [row_lo.byte k, row_hi.byte k], zero extended
}
__device__ __forceinline__ s2v swap2(s2v v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ s2v abs2(s2v v) { return __builtin_elementwise_max(v, (s2v)(-v)); }
__device__ __forceinline__ int hadamard4_rows(const uint32_t c[4], const uint32_t r[4])
{
  s2v m[2][4];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const s2v a = colpair(c[2 * q], c[2 * q + 1], 0) - colpair(r[2 * q], r[2 * q + 1], 0), b = colpair(c[2 * q], c[2 * q + 1], 1) - colpair(r[2 * q], r[2 * q + 1], 1);
    const s2v cc = colpair(c[2 * q], c[2 * q + 1], 2) - colpair(r[2 * q], r[2 * q + 1], 2), e = colpair(c[2 * q], c[2 * q + 1], 3) - colpair(r[2 * q], r[2 * q + 1], 3);
    const s2v s0 = a + e, s1 = b + cc, s2 = b - cc, s3 = a - e;
    m[q][0] = s0 + s1; m[q][1] = s0 - s1; m[q][2] = s2 + s3; m[q][3] = s3 - s2;
  }
  s2v acc = {0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const s2v qs = swap2(m[1][j]);
    const s2v x = abs2(m[0][j] + qs), y = abs2(m[0][j] - qs);                  // |s0|, |s1| and |s3|, |s2|
    acc += __builtin_elementwise_max(x, swap2(x)) + __builtin_elementwise_max(y, swap2(y));
  }
  return (int)(unsigned short)acc.x;
}
__device__ __forceinline__ int sad4_rows(const uint32_t c[4], const uint32_t r[4])
{
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) s = __builtin_amdgcn_sad_u8(c[i], r[i], s);
  return (int)s;
}
__device__ __forceinline__ void had8_1d2(int v[8])
{
  int a[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = v[i] + v[i + 4]; a[i + 4] = v[i] - v[i + 4]; }
  const int b[8] = {a[0] + a[2], a[1] + a[3], a[0] - a[2], a[1] - a[3], a[4] + a[6], a[5] + a[7], a[4] - a[6], a[5] - a[7]};
#pragma unroll
  for (int i = 0; i < 4; i++) { v[2 * i] = b[2 * i] + b[2 * i + 1]; v[2 * i + 1] = b[2 * i] - b[2 * i + 1]; }
}
// HadamardSAD8x8 of the 8x8 block whose current samples start at s_cur row `by`, dword `bx4` and whose reference starts at r
__device__ int hadamard8_lds(const uint32_t *s_cur, int by, int bx4, const uint8_t *r, int pitch)
{
  int m[8][8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t c0 = s_cur[(by + j) * 4 + bx4], c1 = s_cur[(by + j) * 4 + bx4 + 1];
    const uint32_t r0 = ld4(r + (long)j * pitch), r1 = ld4(r + (long)j * pitch + 4);
    int v[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = (int)((c0 >> (8 * i)) & 255) - (int)((r0 >> (8 * i)) & 255); v[4 + i] = (int)((c1 >> (8 * i)) & 255) - (int)((r1 >> (8 * i)) & 255); }
    had8_1d2(v);
#pragma unroll
    for (int i = 0; i < 8; i++) m[j][i] = v[i];
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = m[j][i];
    had8_1d2(v);
#pragma unroll
    for (int j = 0; j < 8; j++) s += iabs2_(v[j]);
  }
  return (s + 2) >> 2;
}

// the nine positions of JM's spiral with R = 1 (mv_search.c:405-442): {0,0} {0,-1} {0,1} {-1,-1} {1,-1} {-1,0} {1,0} {-1,1} {1,1}, two bits per
// coordinate (value + 1) in a constant: the candidate index differs per lane, and a table look-up would be a global load on the path
// from the motion vector to the reference address
__device__ __forceinline__ int sp9_dx(int c) { return (int)((0x22215u >> (2 * c)) & 3u) - 1; }
__device__ __forceinline__ int sp9_dy(int c) { return (int)((0x29421u >> (2 * c)) & 3u) - 1; }
__device__ __constant__ uint8_t c_geom[JMHIP_NPART][4] = {           // x, y, w, h of the 41 partitions (ABI order, jmhip.h)
  {0,0,16,16}, {0,0,16,8},{0,8,16,8}, {0,0,8,16},{8,0,8,16}, {0,0,8,8},{8,0,8,8},{0,8,8,8},{8,8,8,8},
  {0,0,8,4},{8,0,8,4},{0,4,8,4},{8,4,8,4},{0,8,8,4},{8,8,8,4},{0,12,8,4},{8,12,8,4},
  {0,0,4,8},{4,0,4,8},{8,0,4,8},{12,0,4,8},{0,8,4,8},{4,8,4,8},{8,8,4,8},{12,8,4,8},
  {0,0,4,4},{4,0,4,4},{8,0,4,4},{12,0,4,4},{0,4,4,4},{4,4,4,4},{8,4,4,4},{12,4,4,4},
  {0,8,4,4},{4,8,4,4},{8,8,4,4},{12,8,4,4},{0,12,4,4},{4,12,4,4},{8,12,4,4},{12,12,4,4}
};
// partition of block type t+1 that contains 4x4 block (bx4, by4)
__device__ __forceinline__ int part_of(int t, int bx4, int by4)
{
  switch (t) {
  case 0: return 0;
  case 1: return 1 + (by4 >> 1);
  case 2: return 3 + (bx4 >> 1);
  case 3: return 5 + (by4 >> 1) * 2 + (bx4 >> 1);
  case 4: return 9 + by4 * 2 + (bx4 >> 1);
  case 5: return 17 + (by4 >> 1) * 4 + bx4;
  default: return 25 + by4 * 4 + bx4;
  }
}

#define NPAIRS (7 * 16)                             // (block type, 4x4 block) pairs of a macroblock
#define REFINE_THREADS 128
#define COST_MAX 0x7fffffff

// T8MODE: prm.transform8x8_mode != 0 (the Hadamard 8x8 path costs ~64 VGPRs; without it the kernel runs at full occupancy)
template <bool T8MODE>
__global__ __launch_bounds__(REFINE_THREADS, T8MODE ? 3 : 6) void k_me_refine_mb(const jmhip_me_job *__restrict__ jobs, const jmhip_me_result *__restrict__ ires,
                                                                 jmhip_refine_params prm, jmhip_me_result *__restrict__ out,
                                                                 PlaneSet2 ps, const uint8_t *__restrict__ cur, int cur_pitch, int njobs)
{
  __shared__ uint32_t s_cur[64];                  // the current macroblock, 16 rows x 4 dwords
  __shared__ int s_mv[JMHIP_NPART][2];            // per partition: the motion vector the running stage refines
  __shared__ int s_min[JMHIP_NPART];              // min_mcost carried between the stages
  __shared__ unsigned s_dist[JMHIP_NPART * 9];    // distortion of (partition, candidate), summed over its 4x4 / 8x8 blocks
  __shared__ uint8_t s_list[NPAIRS];              // the leader pairs, compacted
  __shared__ int s_nlead[2];
  __shared__ uint16_t s_val[NPAIRS * 9];          // distortion of (block type, 4x4 block, candidate) as its lane computed it, for lanes with the same vector
  if (*ps.jerr) return;                            // a job record failed k_check_me_jobs
  const int tid = threadIdx.x;
  const int jb = xcd_job_index(blockIdx.x, njobs);
  const jmhip_me_job *job = jobs + jb;
  const uint64_t mask = job->part_mask;
  const int mb_x = job->mb_x, mb_y = job->mb_y;
  if (tid < 64) s_cur[tid] = *(const uint32_t *)(cur + (long)(mb_y + (tid >> 2)) * cur_pitch + mb_x + 4 * (tid & 3));
  if (tid < JMHIP_NPART) {
    const jmhip_me_best ib = ires[jb].best[tid];
    s_mv[tid][0] = ib.mv_x; s_mv[tid][1] = ib.mv_y;
    s_min[tid] = prm.start_hp ? ib.cost : COST_MAX;                              // mv_search.c:971-974
  }
  // A lane owns one (block type, 4x4 block) pair for the whole kernel and walks its nine candidates: which partition the block belongs to,
  // where it lies, its current samples and (for SAD) its offset inside the partition are worked out once instead of once per item, and a
  // candidate costs its clamped origin, four loads and the distortion.  112 of the 128 lanes are busy; a lane's nine candidates are
  // independent, so all their reference rows are requested before the first distortion is computed (one memory latency per stage).
  const int pair = tid, t = pair >> 4, b4 = pair & 15, bx4 = b4 & 3, by4 = b4 >> 2;
  const int p = pair < NPAIRS ? part_of(t, bx4, by4) : 0;
  const bool act = pair < NPAIRS && ((mask >> p) & 1);
  const int gx = c_geom[p][0], gy = c_geom[p][1];
#pragma unroll 1
  for (int stage = 0; stage < 2; stage++) {
    const int step = stage == 0 ? 2 : 1, start = stage == 0 ? prm.start_hp : prm.start_qp;
    const int lambda = stage == 0 ? prm.lambda_h : prm.lambda_q, metric = stage == 0 ? prm.metric_h : prm.metric_q;
    for (int k = tid; k < JMHIP_NPART * 9; k += REFINE_THREADS) s_dist[k] = 0;
    __syncthreads();
    const bool sad = metric == JMHIP_METRIC_SAD;
    const bool t8 = T8MODE && p <= 8 && !sad;                                                        // mv_search.c:1630 / :1770
    int leader = -1;
    // Hadamard SATD takes every 4x4 sub-block from its own clamped origin (computeSATD), so the distortion of 4x4 block b at a candidate
    // depends on the block and the vector only, not on the partition it is counted for: when partitions of several block types carry the
    // same vector -- the common case, one motion per macroblock -- the pair of the smallest such type (the leader) is computed, the others
    // pick the nine values up from LDS.  (SAD hangs a partition off one clamped origin, so every pair leads itself; so do the 8x8-Hadamard types.)
    if (act) {
      leader = t;
      if (!sad && !t8) {
        const int mvx = s_mv[p][0], mvy = s_mv[p][1];
        for (int t2 = t - 1; t2 >= 0; t2--) {
          const int p2 = part_of(t2, bx4, by4);
          if (((mask >> p2) & 1) && !(T8MODE && p2 <= 8) && s_mv[p2][0] == mvx && s_mv[p2][1] == mvy) leader = t2;
        }
      }
    }
    // With few leaders the (leader pair, candidate) items are dealt out to ALL lanes -- one motion per macroblock leaves 16 leaders: 144 items
    // are two rounds of the workgroup instead of nine candidates in sequence on 16 lanes.  With many leaders a lane keeps its pair and walks
    // the nine candidates (all 36 rows requested before the first distortion).
    const bool lead4 = act && !t8 && leader == t;
    const unsigned long long lm = __ballot(lead4);
    if ((tid & 63) == 0) s_nlead[tid >> 6] = __popcll(lm);
    __syncthreads();
    const int nlead = s_nlead[0] + s_nlead[1], nc = 9 - start, nitems = nlead * nc;
    const bool spread = nitems <= 4 * REFINE_THREADS;                            // workgroup-uniform
    if (spread) {
      if (lead4) s_list[((tid >> 6) ? s_nlead[0] : 0) + __popcll(lm & ((1ull << (tid & 63)) - 1))] = (uint8_t)pair;
      __syncthreads();
#pragma unroll 1
      for (int it0 = tid; it0 < nitems; it0 += 2 * REFINE_THREADS) {
        const int it1 = it0 + REFINE_THREADS;
        const bool two = it1 < nitems;
        int prs[2], cands[2];
        uint32_t c[2][4], rr[2][4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int it = (h == 0 || two) ? (h ? it1 : it0) : it0;                // the idle half repeats item 0 (its result is dropped)
          const int li = it / nc, cand = start + (it - li * nc), pr = s_list[li];
          const int tt = pr >> 4, bb = pr & 15, bx = bb & 3, by = bb >> 2, pp = part_of(tt, bx, by);
          const int ggx = c_geom[pp][0], ggy = c_geom[pp][1];
          const int qx0 = ((sad ? mb_x + ggx : mb_x + 4 * bx) << 2) + s_mv[pp][0], qy0 = ((sad ? mb_y + ggy : mb_y + 4 * by) << 2) + s_mv[pp][1];
          const uint32_t inner = sad ? (uint32_t)(__mul24(4 * by - ggy, ps.pitch) + (4 * bx - ggx)) : 0u, pw = (uint32_t)ps.pitch;
          const uint32_t r = umv_off2(ps, qy0 + sp9_dy(cand) * step, qx0 + sp9_dx(cand) * step) + inner;
          rr[h][0] = ld4o(ps.base, r); rr[h][1] = ld4o(ps.base, r + pw); rr[h][2] = ld4o(ps.base, r + 2 * pw); rr[h][3] = ld4o(ps.base, r + 3 * pw);
#pragma unroll
          for (int k = 0; k < 4; k++) c[h][k] = s_cur[(4 * by + k) * 4 + bx];
          prs[h] = pr; cands[h] = cand;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const unsigned d = (unsigned)(sad ? sad4_rows(c[h], rr[h]) : hadamard4_rows(c[h], rr[h]));
          if (h == 0 || two) s_val[prs[h] * 9 + cands[h]] = (uint16_t)d;
        }
      }
    } else if (act && !t8) {
      if (leader == t) {
        const int mvx = s_mv[p][0], mvy = s_mv[p][1];
        // SAD: the whole partition hangs off ONE clamped origin (computeSAD), this block sits at a fixed offset from it;
        // SATD: every 4x4 (8x8) sub-block has its own clamped origin (computeSATD)
        const int qx0 = ((sad ? mb_x + gx : mb_x + 4 * bx4) << 2) + mvx, qy0 = ((sad ? mb_y + gy : mb_y + 4 * by4) << 2) + mvy;
        const uint32_t inner = sad ? (uint32_t)(__mul24(4 * by4 - gy, ps.pitch) + (4 * bx4 - gx)) : 0u, pw = (uint32_t)ps.pitch;
        const uint32_t c[4] = {s_cur[(4 * by4) * 4 + bx4], s_cur[(4 * by4 + 1) * 4 + bx4], s_cur[(4 * by4 + 2) * 4 + bx4], s_cur[(4 * by4 + 3) * 4 + bx4]};
        // three candidates at a time (12 rows in flight): this path runs when most pairs lead themselves, and the registers of all nine
        // candidates' rows would halve the number of workgroups a compute unit holds
#pragma unroll
        for (int c0 = 0; c0 < 9; c0 += 3) {
          uint32_t rr[3][4];
#pragma unroll
          for (int k = 0; k < 3; k++)
            if (c0 + k >= start) {
              const uint32_t r = umv_off2(ps, qy0 + sp9_dy(c0 + k) * step, qx0 + sp9_dx(c0 + k) * step) + inner;
              rr[k][0] = ld4o(ps.base, r); rr[k][1] = ld4o(ps.base, r + pw); rr[k][2] = ld4o(ps.base, r + 2 * pw); rr[k][3] = ld4o(ps.base, r + 3 * pw);
            }
#pragma unroll
          for (int k = 0; k < 3; k++)
            if (c0 + k >= start) {
              const unsigned d = (unsigned)(sad ? sad4_rows(c, rr[k]) : hadamard4_rows(c, rr[k]));
              s_val[pair * 9 + c0 + k] = (uint16_t)d;
              atomicAdd(&s_dist[p * 9 + c0 + k], d);
            }
        }
      }
    }
    if (T8MODE && act && t8 && !((bx4 | by4) & 1)) {                              // the top-left 4x4 leads its 8x8 block
      const int qx0 = ((mb_x + 4 * bx4) << 2) + s_mv[p][0], qy0 = ((mb_y + 4 * by4) << 2) + s_mv[p][1];
#pragma unroll 1
      for (int cand = start; cand < 9; cand++) {
        const uint32_t r = umv_off2(ps, qy0 + sp9_dy(cand) * step, qx0 + sp9_dx(cand) * step);
        atomicAdd(&s_dist[p * 9 + cand], (unsigned)hadamard8_lds(s_cur, 4 * by4, bx4, ps.base + (size_t)r, ps.pitch));
      }
    }
    // values computed for another pair (or, after the items were dealt out, for any pair) are added up here
    const bool pick = act && !t8 && (spread || leader != t);
    if (__syncthreads_or(pick)) {                             // workgroup-uniform
      if (pick)
        for (int cand = start; cand < 9; cand++) atomicAdd(&s_dist[p * 9 + cand], (unsigned)s_val[(leader * 16 + b4) * 9 + cand]);
      __syncthreads();
    }
    if (tid < JMHIP_NPART && ((mask >> tid) & 1)) {
      const int p = tid, mvx = s_mv[p][0], mvy = s_mv[p][1], pred_x = job->pred[p][0], pred_y = job->pred[p][1];
      int min_mcost = (stage == 1 && !prm.start_qp) ? COST_MAX : s_min[p];     // me_fullsearch.c:252-253
      int best = 0;
      const int carried = min_mcost;
#pragma unroll
      for (int l = 0; l < 9; l++) {
        int cost;
        if (l < start) cost = l == 0 ? carried : COST_MAX;                      // position 0 keeps the carried-in cost
        else {
          const int cx = mvx + sp9_dx(l) * step, cy = mvy + sp9_dy(l) * step;
          cost = lambda * (mvbits2(cx - pred_x) + mvbits2(cy - pred_y)) + (int)(s_dist[p * 9 + l] << 5);
        }
        if (cost < min_mcost) { min_mcost = cost; best = l; }
      }
      s_mv[p][0] = mvx + sp9_dx(best) * step; s_mv[p][1] = mvy + sp9_dy(best) * step;
      s_min[p] = min_mcost;
    }
    __syncthreads();
  }
  if (tid < JMHIP_NPART && ((mask >> tid) & 1)) {
    jmhip_me_best b;
    b.mv_x = (int16_t)s_mv[tid][0]; b.mv_y = (int16_t)s_mv[tid][1]; b.cost = s_min[tid];
    out[jb].best[tid] = b;
  }
}

void jmhip_launch_refine_mb(jmhip_ctx *ctx, int slot, const jmhip_me_job *d_jobs, int njobs, const jmhip_me_result *d_int,
                            const jmhip_refine_params *prm, jmhip_me_result *d_out)
{
  PlaneSet2 ps; ps.base = ctx->d_sub[slot]; ps.pitch = ctx->pitch; ps.plane_stride = (long)ctx->plane_stride; ps.W = ctx->W; ps.H = ctx->H; ps.jerr = ctx->d_me_declined + 4;
  if (prm->transform8x8_mode)
    hipLaunchKernelGGL(k_me_refine_mb<true>, dim3(njobs), dim3(REFINE_THREADS), 0, ctx->stream, d_jobs, d_int, *prm, d_out, ps, ctx->d_cur, ctx->cur_pitch, njobs);
  else
    hipLaunchKernelGGL(k_me_refine_mb<false>, dim3(njobs), dim3(REFINE_THREADS), 0, ctx->stream, d_jobs, d_int, *prm, d_out, ps, ctx->d_cur, ctx->cur_pitch, njobs);
}
