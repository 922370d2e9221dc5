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

static __device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
static __device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static __device__ __forceinline__ bool is_intra(int t) { return t == 9 || t == 13 || t == 10 || t == 14; }
static __device__ __forceinline__ int cmp_mv(const int16_t a[2], const int16_t b[2]) { return (int)(iabs_(a[0] - b[0]) >= 4) | (int)(iabs_(a[1] - b[1]) >= 4); }

// LDS tile: luma 20 x 20 (rows/cols -4..15), pitch 24; chroma up to 18 x 10 (rows -2..15, cols -2..7), pitch 12
#define LP 24
#define CP 12

static __device__ void luma_line(uint8_t *q0p, int st, int bS, int alpha, int beta, int c0)
{
  const int L0 = q0p[-st], R0 = q0p[0];
  if (bS == 4) {
    if (iabs_(R0 - L0) < alpha) {
      const int R1 = q0p[st], L1 = q0p[-2 * st];
      if (iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
        const int R2 = q0p[2 * st], L2 = q0p[-3 * st], RL0 = L0 + R0;
        const int small_gap = iabs_(R0 - L0) < ((alpha >> 2) + 2);
        const int aq = (iabs_(R0 - R2) < beta) & small_gap, ap = (iabs_(L0 - L2) < beta) & small_gap;
        if (ap) {
          const int L3 = q0p[-4 * st];
          q0p[-st]     = (uint8_t)((R1 + ((L1 + RL0) << 1) + L2 + 4) >> 3);
          q0p[-2 * st] = (uint8_t)((L2 + L1 + RL0 + 2) >> 2);
          q0p[-3 * st] = (uint8_t)((((L3 + L2) << 1) + L2 + L1 + RL0 + 4) >> 3);
        } else q0p[-st] = (uint8_t)(((L1 << 1) + L0 + R1 + 2) >> 2);
        if (aq) {
          const int R3 = q0p[3 * st];
          q0p[0]      = (uint8_t)((L1 + ((R1 + RL0) << 1) + R2 + 4) >> 3);
          q0p[st]     = (uint8_t)((R2 + R0 + L0 + R1 + 2) >> 2);
          q0p[2 * st] = (uint8_t)((((R3 + R2) << 1) + R2 + R1 + RL0 + 4) >> 3);
        } else q0p[0] = (uint8_t)(((R1 << 1) + R0 + L1 + 2) >> 2);
      }
    }
  } else if (bS != 0) {
    const int diff = R0 - L0;
    if (iabs_(diff) < alpha) {
      const int R1 = q0p[st], L1 = q0p[-2 * st];
      if (iabs_(R0 - R1) < beta && iabs_(L0 - L1) < beta) {
        const int RL0 = (L0 + R0 + 1) >> 1, R2 = q0p[2 * st], L2 = q0p[-3 * st];
        const int aq = iabs_(R0 - R2) < beta, ap = iabs_(L0 - L2) < beta;
        const int tc = c0 + ap + aq;
        const int dif = clip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
        if (ap) q0p[-2 * st] = (uint8_t)(L1 + clip3(-c0, c0, (L2 + RL0 - (L1 << 1)) >> 1));
        if (dif != 0) { q0p[-st] = (uint8_t)clip3(0, 255, L0 + dif); q0p[0] = (uint8_t)clip3(0, 255, R0 - dif); }
        if (aq) q0p[st] = (uint8_t)(R1 + clip3(-c0, c0, (R2 + RL0 - (R1 << 1)) >> 1));
      }
    }
  }
}

static __device__ void chroma_line(uint8_t *q0p, int st, int bS, int alpha, int beta, int c0)
{
  const int L0 = q0p[-st], R0 = q0p[0], diff = R0 - L0;
  if (bS == 0 || iabs_(diff) >= alpha) return;
  const int R1 = q0p[st], L1 = q0p[-2 * st];
  if (iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
  if (bS == 4) {
    q0p[-st] = (uint8_t)(((L1 << 1) + L0 + R1 + 2) >> 2);
    q0p[0]   = (uint8_t)(((R1 << 1) + R0 + L1 + 2) >> 2);
  } else {
    const int tc = c0 + 1, dif = clip3(-tc, tc, ((diff << 2) + (L1 - R1) + 4) >> 3);
    if (dif != 0) { q0p[-st] = (uint8_t)clip3(0, 255, L0 + dif); q0p[0] = (uint8_t)clip3(0, 255, R0 - dif); }
  }
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

