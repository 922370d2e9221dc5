// deblock_common.h -- tables, edge filters and boundary-strength derivation shared by the deblocking kernels
// (deblock.hip: one launch per 2:1 diagonal; deblock_rows.hip: row pipeline).  Reference: lencod/inc/loop_filter.h:32-59,
// lencod/src/loop_filter_normal.c:52-292 (strengths), :301-757 (edge filters).
#pragma once
#include "jmhip_internal.h"

static __device__ __constant__ uint8_t c_alpha[52] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,
                                               32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255};
static __device__ __constant__ uint8_t c_beta[52]  = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,
                                               9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18};
static __device__ __constant__ uint8_t c_tc0[52][4] = {   // CLIP_TAB columns bS = 0..3
  {0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},
  {0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,1},{0,0,0,1},{0,0,0,1},{0,0,0,1},{0,0,1,1},{0,0,1,1},{0,1,1,1},{0,1,1,1},{0,1,1,1},
  {0,1,1,1},{0,1,1,2},{0,1,1,2},{0,1,1,2},{0,1,1,2},{0,1,2,3},{0,1,2,3},{0,2,2,3},{0,2,2,4},{0,2,3,4},{0,2,3,4},{0,3,3,5},{0,3,4,6},
  {0,3,4,6},{0,4,5,7},{0,4,5,8},{0,4,6,9},{0,5,7,10},{0,6,8,11},{0,6,8,13},{0,7,10,14},{0,8,11,16},{0,9,12,18},{0,10,13,20},
  {0,11,15,23},{0,13,17,25}
};
static __device__ __constant__ int8_t c_chroma_edge[2][4][4] = {
  {{-4, 0, 0, 0}, {-4, -4, -4, 4}, {-4, 4, 4, 8}, {-4, -4, -4, 12}},
  {{-4, 0, 0, 0}, {-4, -4, 4, 4},  {-4, 4, 8, 8}, {-4, -4, 12, 12}}
};

#ifndef JMHIP_HAVE_IABS                                         // me_common.h has the same (mbpipe.hip includes both)
static __device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
#endif
static __device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static __device__ __forceinline__ bool is_intra(int t) { return t == 9 || t == 13 || t == 10 || t == 14; }
static __device__ __forceinline__ int cmp_mv(const int16_t a[2], const int16_t b[2]) { return (int)(iabs_(a[0] - b[0]) >= 4) | (int)(iabs_(a[1] - b[1]) >= 4); }

// LDS tile: luma 20 x 20 (rows/cols -4..15), pitch 24; chroma up to 18 x 10 (rows -2..15, cols -2..7), pitch 12
#define LP 24
#define CP 12

// One line of samples across an edge, H.264 8.7.2.3 / 8.7.2.4 in the standard's own notation (p3 p2 p1 p0 | q0 q1 q2 q3), used by the
// one-launch-per-diagonal fallback (k_deblock_diag: planes whose alignment the band kernels cannot take).  All eight samples are read, the six
// results are formed as selects and stored where they differ -- the same arithmetic as the band kernels' filters (deblock_rows.hip), one line at a time.
static __device__ void filter_luma_line(uint8_t *e, int st, int bS, int alpha, int beta, int tc0)
{
  if (bS == 0) return;
  int p[4], q[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { p[k] = e[-(k + 1) * st]; q[k] = e[k * st]; }
  const int d0 = q[0] - p[0];
  if (!(iabs_(d0) < alpha && iabs_(p[1] - p[0]) < beta && iabs_(q[1] - q[0]) < beta)) return;        // filterSamplesFlag (8-468)
  const bool ap = iabs_(p[2] - p[0]) < beta, aq = iabs_(q[2] - q[0]) < beta;
  int np[3] = {p[0], p[1], p[2]}, nq[3] = {q[0], q[1], q[2]};
  if (bS < 4) {                                                 // 8.7.2.3
    const int tc = tc0 + (int)ap + (int)aq;
    const int delta = clip3(-tc, tc, ((d0 << 2) + (p[1] - q[1]) + 4) >> 3);
    const int mid = (p[0] + q[0] + 1) >> 1;
    np[0] = clip3(0, 255, p[0] + delta); nq[0] = clip3(0, 255, q[0] - delta);
    if (ap) np[1] = p[1] + clip3(-tc0, tc0, (p[2] + mid - (p[1] << 1)) >> 1);
    if (aq) nq[1] = q[1] + clip3(-tc0, tc0, (q[2] + mid - (q[1] << 1)) >> 1);
  } else {                                                      // 8.7.2.4
    const bool flat = iabs_(d0) < ((alpha >> 2) + 2);
    const int s = p[0] + q[0];
    if (ap && flat) {
      np[0] = (p[2] + 2 * p[1] + 2 * s + q[1] + 4) >> 3;
      np[1] = (p[2] + p[1] + s + 2) >> 2;
      np[2] = (2 * p[3] + 3 * p[2] + p[1] + s + 4) >> 3;
    } else np[0] = (2 * p[1] + p[0] + q[1] + 2) >> 2;
    if (aq && flat) {
      nq[0] = (q[2] + 2 * q[1] + 2 * s + p[1] + 4) >> 3;
      nq[1] = (q[2] + q[1] + s + 2) >> 2;
      nq[2] = (2 * q[3] + 3 * q[2] + q[1] + s + 4) >> 3;
    } else nq[0] = (2 * q[1] + q[0] + p[1] + 2) >> 2;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (np[k] != p[k]) e[-(k + 1) * st] = (uint8_t)np[k];
    if (nq[k] != q[k]) e[k * st] = (uint8_t)nq[k];
  }
}

static __device__ void filter_chroma_line(uint8_t *e, int st, int bS, int alpha, int beta, int tc0)
{
  if (bS == 0) return;
  const int p0 = e[-st], p1 = e[-2 * st], q0 = e[0], q1 = e[st], d0 = q0 - p0;
  if (!(iabs_(d0) < alpha && iabs_(p1 - p0) < beta && iabs_(q1 - q0) < beta)) return;
  int n0, m0;
  if (bS < 4) {
    const int tc = tc0 + 1, delta = clip3(-tc, tc, ((d0 << 2) + (p1 - q1) + 4) >> 3);
    n0 = clip3(0, 255, p0 + delta); m0 = clip3(0, 255, q0 - delta);
  } else { n0 = (2 * p1 + p0 + q1 + 2) >> 2; m0 = (2 * q1 + q0 + p1 + 2) >> 2; }
  e[-st] = (uint8_t)n0; e[0] = (uint8_t)m0;
}

// boundary strength of one 4-sample segment (GetStrengthVer/Hor); dir 0 = vertical edge
static __device__ int strength_of(int dir, int edge, int idx, int addr, int mb_w, const jmhip_db_mb *mbs, const jmhip_db_motion *motion)
{
  const jmhip_db_mb *q = &mbs[addr];
  if (q->slice_type == 3 || q->slice_type == 4) return edge == 0 ? 4 : 3;
  const jmhip_db_mb *p = edge ? q : (dir == 0 ? &mbs[addr - 1] : &mbs[addr - mb_w]);
  if (is_intra(q->mb_type) || is_intra(p->mb_type)) return edge == 0 ? 4 : 3;
  int qbx, qby, pbx, pby;
  if (dir == 0) { qbx = edge; qby = idx; pbx = (edge + 3) & 3; pby = idx; }
  else          { qbx = idx; qby = edge; pbx = idx; pby = (edge + 3) & 3; }
  if (((q->cbp_blk >> (qby * 4 + qbx)) & 1) || ((p->cbp_blk >> (pby * 4 + pbx)) & 1)) return 2;
  if (edge && (q->mb_type == 1 || q->mb_type == (dir == 0 ? 2 : 3))) return 0;
  const int mbx = addr % mb_w, mby = addr / mb_w, bw = mb_w * 4;
  const int qx = mbx * 4 + qbx, qy = mby * 4 + qby, px = dir == 0 ? qx - 1 : qx, py = dir == 0 ? qy : qy - 1;
  const jmhip_db_motion *a = &motion[qy * bw + qx], *b = &motion[py * bw + px];
  const int a0 = a->ref_id[0], a1 = a->ref_id[1], b0 = b->ref_id[0], b1 = b->ref_id[1];
  if ((a0 == b0 && a1 == b1) || (a0 == b1 && a1 == b0)) {
    if (a0 != a1) {
      if (a0 == b0) return cmp_mv(a->mv[0], b->mv[0]) | cmp_mv(a->mv[1], b->mv[1]);
      return cmp_mv(a->mv[0], b->mv[1]) | cmp_mv(a->mv[1], b->mv[0]);
    }
    return (cmp_mv(a->mv[0], b->mv[0]) | cmp_mv(a->mv[1], b->mv[1])) && (cmp_mv(a->mv[0], b->mv[1]) | cmp_mv(a->mv[1], b->mv[0]));
  }
  return 1;
}

